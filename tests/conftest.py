import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # torch bundles its own ROCm runtime: it has to initialise BEFORE libpt_hip.so pulls in /opt/rocm's, otherwise
    # torch later reports "No HIP GPUs are available" (the tile tests use torch tensors as device buffers).
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """CPU-side native pieces (host mirror, oracle, host-compiled device headers) are built on demand; the HIP
    library is built by __graft_entry__.build() (it travels to the GPU box prebuilt)."""
    import __graft_entry__ as g

    g.build_host()
    g.build_oracle()
    g.build_test_shim()
    if not os.path.exists(os.path.join(g.PKG, "libpt_hip.so")):
        g.build_hip()


@pytest.fixture(scope="session")
def dxrs():
    import dxrs_amd_loader  # noqa: F401
    import dxrs_amd

    return dxrs_amd


@pytest.fixture(scope="session")
def host(dxrs):
    return dxrs.load_host()


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def renderer(dxrs):
    """One GPU context shared by the gpu tests (single process, single context)."""
    r = dxrs.Renderer(device=0)
    yield r
    r.close()


@pytest.fixture(scope="session")
def renderer_no_beams(dxrs):
    """A context created with PT_BEAMS=0 (the knobs are read once, at pt_create): every primary ray traverses the BVH."""
    old = os.environ.get("PT_BEAMS")
    os.environ["PT_BEAMS"] = "0"
    try:
        r = dxrs.Renderer(device=0)
    finally:
        if old is None:
            os.environ.pop("PT_BEAMS", None)
        else:
            os.environ["PT_BEAMS"] = old
    yield r
    r.close()
