"""PCD DAG branches on several GPUs (BASELINE configs[4]): independent `ECCyclePCD::prove` calls
(/root/reference src/ec_cycle_pcd/mod.rs:92-181 -- the per-prior loop of a merge node, data_structures.rs:269-304, consumes
proofs that were made independently) map one per device with no exchange at all: one host thread + one ordinary pcdhip context per
branch, N threads x N contexts.  ctypes releases the GIL inside every library call, so the threads overlap like the reference's
host threads would.  The merge node itself then proves through ONE multi-device context (capi.Context(devices=[...])) whose MSMs
use all GPUs."""
import threading

from . import capi


def run_branches(branches, devices):
    """branches: list of callables `f(ctx) -> result`; branch i runs on devices[i % len(devices)] in its own host thread with its
    own context.  Returns the results in order; the first exception of any branch is re-raised."""
    results = [None] * len(branches)
    errors = [None] * len(branches)

    def work(i, fn, dev):
        try:
            ctx = capi.Context(dev)
            try:
                results[i] = fn(ctx)
            finally:
                ctx.close()
        except BaseException as e:  # noqa: BLE001 -- handed to the caller below
            errors[i] = e

    threads = [threading.Thread(target=work, args=(i, fn, devices[i % len(devices)])) for i, fn in enumerate(branches)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    return results
