"""Data-parallel step on the GPU: 2 ranks (sharing the box's single GPU over gloo) == 1 rank on the global batch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env, tag):
    """The worker ranks write their error streams to files of their own (C2_DDP_LOG_DIR): shown when the run fails."""
    logs = os.path.join(ROOT, "gpurun_out", "ddp_logs", tag)
    r = subprocess.run(cmd, env=dict(env, C2_DDP_LOG_DIR=logs), capture_output=True, text=True, timeout=110)
    if r.returncode != 0:
        per_rank = ""
        for f in sorted(os.listdir(logs)) if os.path.isdir(logs) else []:
            per_rank += f"\n--- {f}\n" + open(os.path.join(logs, f)).read()[-3000:]
        raise AssertionError(r.stdout[-3000:] + r.stderr[:1500] + "\n...\n" + r.stderr[-3000:] + per_rank)


@pytest.mark.parametrize("variant", ["ln", "bn", "noise", "shard", "shard_bf16"])
def test_two_ranks_equal_single_process(variant):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "tests", "ddp_worker.py"), variant]
    _run(cmd, env, "gloo2_" + variant)


@pytest.mark.parametrize("variant", ["ln", "bn", "shard", "shard_bf16"])
def test_one_rank_over_rccl_issues_every_collective(variant):
    """The RCCL branch of parallel.py on the one GPU of the box: a one-rank "nccl" group with C2_DP_SINGLE_RANK=1 still
    all-gathers the latent codes / BatchNorm partials, reduces (or reduce-scatters) every gradient bucket on the side
    stream with the last one deferred into the optimizer, and all-gathers the parameter chunks -- real RCCL calls,
    stream-ordered, on device tensors.  The result must equal the plain single-process step."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", C2_DDP_BACKEND="nccl", C2_DP_SINGLE_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(29950 + os.getpid() % 40), os.path.join(ROOT, "tests", "ddp_worker.py"), variant]
    _run(cmd, env, "rccl1_" + variant)


@pytest.mark.parametrize("variant", ["allreduce", "shard", "shard_bf16", "bn"])
def test_data_parallel_step_replayed_as_a_graph(variant):
    """Trainer(graph=True, world=...): the data-parallel iteration INCLUDING its RCCL collectives captured once and replayed
    (one rank over RCCL issuing every collective, as above) == the same iteration launched kernel by kernel, bit for bit
    (weights, optimizer step counts, logs).  Reference step being sharded: trainer.py:305-323."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", C2_DP_SINGLE_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(29860 + os.getpid() % 40), os.path.join(ROOT, "tests", "dp_graph_worker.py"), variant]
    _run(cmd, env, "dpgraph_" + variant)
