"""Child process of tests/test_gpu_parity.py::test_arena_allocation_failure_falls_back_to_uniform_buffers: runs with
RABITQ_HIP_SO pointing at the DEVELOPER build (librabitq_hip_dev.so, `make dev`, -DRQ_DEV_ABLATIONS), the only library that carries
the allocation-failure injection (option survivor_segments = 3: every arena stage fails through a real, oversized hipMalloc).  The
pass must be repeated on the uniform survivor buffers with results bit-identical to the oracle, for both scan implementations."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def arena_failure():
    import oracle
    import rabitq_amd as rq
    from rabitq_amd import _lib, index as ix
    from tests import synth
    from tests.test_gpu_parity import _compare_with_oracle
    assert _lib.SO_PATH.endswith("librabitq_hip_dev.so"), _lib.SO_PATH
    oracle.build()
    _lib.check(_lib.lib().rq_init(0))
    n, d, k = 300_000, 64, 6          # the workload of test_segmented_final_stage_matches_oracle
    rng = np.random.default_rng(33)
    centres = (rng.standard_normal((k, d)) * 4.0).astype(np.float32)
    sizes = np.array([0.8, 0.04, 0.04, 0.04, 0.04, 0.04])
    lab = rng.choice(k, n, p=sizes)
    x = (centres[lab] + rng.standard_normal((n, d))).astype(np.float32)
    P = synth.random_orthogonal(d, seed=34)
    oidx = oracle.OracleIndex.build(x, centres, P)
    nq = 300
    queries = (centres[rng.choice(k, nq, p=sizes)] + 0.2 * rng.standard_normal((nq, d))).astype(np.float32)
    queries[:40] = (x[lab == 0][:40] + 0.1 * rng.standard_normal((40, d))).astype(np.float32)
    for impl in (1, 2):
        ix.set_option("scan_impl", impl)
        ix.set_option("survivor_segments", 2)
        gidx = rq.RaBitQ.build(x, centres, P)
        _compare_with_oracle(rq, oracle, oidx, gidx, queries, k, 100, False)
        assert ix.last_profile()["segmented_passes"] == 1
        ix.set_option("survivor_segments", 3)        # every arena stage fails: the pass is repeated on the uniform buffers
        _compare_with_oracle(rq, oracle, oidx, gidx, queries, k, 100, False)
        assert ix.last_profile()["segmented_passes"] == 0
        _compare_with_oracle(rq, oracle, oidx, gidx, queries, 3, 10, True)
        ix.set_option("survivor_segments", 2)        # ... and the arena is back afterwards
        _compare_with_oracle(rq, oracle, oidx, gidx, queries, k, 100, False)
        assert ix.last_profile()["segmented_passes"] == 1
        gidx.close()
    oidx.close()
    print("DEV_HOOK_OK")


if __name__ == "__main__":
    {"arena_failure": arena_failure}[sys.argv[1]]()
