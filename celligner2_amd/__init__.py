"""celligner2_amd: MI355X-native training step of Celligner2's conditional VAE."""
