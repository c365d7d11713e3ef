import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--gemm-mode", default=None, choices=["f32", "bf16x3", "bf16"],
                     help="run the whole suite with this GEMM arithmetic mode (library-wide nacf_gemm_set_mode); "
                          "default: the library's own default")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _gemm_mode(request):
    mode = request.config.getoption("--gemm-mode")
    if mode is not None:
        import nacf_amd  # noqa: F401
        from nacf_amd.runtime import ops
        ops.set_gemm_mode(mode)
    yield


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# tuning / test knobs the library reads with getenv() on every call: a test may set them ONLY through monkeypatch
# (restored afterwards).  A leaked value would silently change which kernels every later test exercises -- e.g. the
# golden train / decode parity of test_model_gpu.py running on forced 128x128 tiles and not on the production
# heuristic -- so every test starts by asserting that none of them is set.
TUNING_ENV = ("NACF_DDP_STAGES", "NACF_GEMM_TILE", "NACF_GEMM_SPLITS", "NACF_GEMM_MODE", "NACF_ATTN_VALU", "NACF_ATTN_LDS", "NACF_ATTN_KB", "NACF_ATTN_BF16",
              "NACF_ATTN_WPI", "NACF_HIP_LIB", "NACF_GEMM_GROUP_N", "NACF_DW_GROUP", "NACF_DW_GROUP_WGS", "NACF_GEMM_WIDE",
              "NACF_FUSED_ZERO_GRAD", "NACF_BN_MULTI", "NACF_SYNC_BN_EXCHANGES", "NACF_DDP_GRAPH_COLLECTIVES", "NACF_DW_GROUP_ORDER", "NACF_DEAD_ROWS", "NACF_SELFTEST")


@pytest.fixture(autouse=True)
def _no_leaked_tuning_env():
    leaked = [k for k in TUNING_ENV if k in os.environ]
    assert not leaked, "tuning environment leaked into this test: %s" % leaked
    yield
