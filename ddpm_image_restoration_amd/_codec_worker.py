"""Host codec worker process: `python -m ddpm_image_restoration_amd._codec_worker <shm_in> <shm_out>`.

Pillow's AVIF encoder does not run concurrently from threads of one process (measured on the MI355X host: 16 images
take 591 ms sequentially and 454 ms from 16 threads), so the AVIF degradation operator of the sampling loop
(avif_inference.py:64-98) is spread over worker PROCESSES instead.  A worker imports numpy and Pillow only (never
torch / HIP), attaches two shared-memory blocks holding the uint8 NCHW batch in and out, and serves one-line
requests on stdin: "<index> <H> <W> <quality> <kind>" -> "ok" / "err <message>" on stdout.
"""
from __future__ import annotations

import sys
from multiprocessing import shared_memory

import numpy as np


def main(argv) -> int:
    from . import codec
    shm_in = shared_memory.SharedMemory(name=argv[0])
    shm_out = shared_memory.SharedMemory(name=argv[1])
    # Python < 3.13 registers ATTACHED blocks with this process's resource tracker, which would unlink the parent's
    # blocks when the worker exits: the parent owns them
    from multiprocessing import resource_tracker
    for shm in (shm_in, shm_out):
        try:
            resource_tracker.unregister(shm._name, "shared_memory")
        except Exception:
            pass
    try:
        for line in sys.stdin:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "quit":
                break
            try:
                idx, h, w, q = (int(v) for v in parts[:4])
                kind = parts[4]
                n = 3 * h * w
                src = np.ndarray((3, h, w), dtype=np.uint8, buffer=shm_in.buf, offset=idx * n)
                dst = np.ndarray((3, h, w), dtype=np.uint8, buffer=shm_out.buf, offset=idx * n)
                dst[...] = codec._one(src, q, kind)
                sys.stdout.write("ok\n")
            except Exception as e:  # reported to the parent, which raises
                sys.stdout.write("err " + repr(e).replace("\n", " ") + "\n")
            sys.stdout.flush()
    finally:
        shm_in.close()
        shm_out.close()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
