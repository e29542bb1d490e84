import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    try:
        from cdpr_simulation_amd._native import lib

        return lib().cdpr_device_count()
    except Exception:
        return 0


def pytest_sessionstart(session):
    """The native pieces are built in-tree and travel with the snapshot; if a checkout arrives without them, build
    them once (hipcc cross-compiles without a GPU).  A missing library afterwards still fails loudly in the tests."""
    lib = os.path.join(ROOT, "cdpr-simulation_amd", "libcdpr_hip.so")
    ora = os.path.join(ROOT, "oracle", "libcdpr_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        try:
            import __graft_entry__

            __graft_entry__.build()
        except Exception as exc:  # noqa: BLE001
            print(f"[conftest] native build failed: {exc}")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never pass silently without the HIP library: they are only
    # deselected by `-m "not gpu"`; on a box with no GPU they fail at cdpr_create.
    pass


@pytest.fixture(scope="session")
def pkg():
    import cdpr_simulation_amd

    return cdpr_simulation_amd


@pytest.fixture(scope="session")
def oracle():
    import oracle as _oracle

    _oracle.build()
    return _oracle


def mapped_words(n):
    """uint32[n] in pinned host memory mapped to the device (fine-grained): (numpy view, device pointer, free).  What a HOST producer
    posts a schedule's mailbox through (include/cdpr.h: cdpr_update_scheduled, d_ready): plain stores, no GPU queue in between - a copy
    enqueued on a stream can land on the hardware queue of the launch that is waiting for it and never complete."""
    import ctypes as C

    import numpy as np

    try:
        hip = C.CDLL("libamdhip64.so")
    except OSError:
        hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    host, dev = C.c_void_p(), C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(host), C.c_size_t(4 * n), C.c_uint(0x2 | 0x40000000))  # hipHostMallocMapped | hipHostMallocCoherent
    assert rc == 0, rc
    rc = hip.hipHostGetDevicePointer(C.byref(dev), host, C.c_uint(0))
    assert rc == 0, rc
    words = np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_uint32)), shape=(n,))
    words[:] = 0
    return words, int(dev.value), lambda: hip.hipHostFree(host)
