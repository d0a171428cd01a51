import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rxinfer.jl_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# the schedule switches the tests flip (RXHIP_GSEQ, RXHIP_ONE_PASS, … — include/rxhip.h "Environment") are read by the library only with this set
os.environ["RXHIP_TEST_HOOKS"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a device: skip (not fail) them on a box without one."""
    try:
        import rxhip

        have = rxhip.lib().rxhip_device_count() > 0
    except Exception:  # library not built: the non-gpu tests report that loudly (tests/test_abi.py)
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The CPU oracle is compiled on demand; librxhip.so must already exist (build())."""
    import rxoracle

    rxoracle.build()
    yield
