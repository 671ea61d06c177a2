import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_step_gpu as T
for hidden in ([508, 256], [516, 256], [524, 256], [492, 256], [820, 512], [510, 256]):
    print(hidden, flush=True)
    try:
        T.test_bf16_step_midsize_vs_oracle(False, hidden)
    except AssertionError as e:
        print("  FAILED")
