#!/usr/bin/env python3
"""the HBM-resident streamed kernel on configs[4] (5000 x 10000, min-sum, 32768 frames, 50 sweeps) was seen at
158 ms and at 179 ms.  One process, the same kernel instance, several allocation histories: which of them moves the time?

  fresh        decoder created first thing in the process
  after_free   after a large torch allocation was made and released (caching allocator emptied)
  after_hold   while 64 / 128 GB of other device memory stay allocated (the slab workspace lands elsewhere)
  recreate     the same again, new handle each time (the workspace is a new hipMalloc every time)
Every line: workspace base address (and its offset inside a 2 MiB / 1 GiB frame), per-launch kernel ms.
"""
import os
import re
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import acg_alp_ldpc_amd as A
    from acg_alp_ldpc_amd._lib import McCfg, check, lib
    import ctypes as C
    F = 32768
    Hm = A.regular_ldpc(5000, 10000, 3, 6, seed=1)
    H = A.ParityCheckMatrix(Hm)
    n, nw = H.n, (H.n + 31) // 32
    stream = torch.cuda.Stream()
    cws = np.zeros((1, n), dtype=np.uint8)

    def run(tag, launches=4):
        dec = A.MinSumDecoder(50, 0.75, early_exit=False, engine=A.ENGINE_STREAMED)
        desc = dec.describe(H)
        base = int(re.search(r"workspace_base=(0x[0-9a-f]+)", desc).group(1), 16)
        y = torch.empty((F, n), dtype=torch.float32, device="cuda")
        bits = torch.zeros((F, nw), dtype=torch.int32, device="cuda")
        ok = torch.zeros(F, dtype=torch.uint8, device="cuda")
        its = torch.zeros(F, dtype=torch.int32, device="cuda")
        h, _ = dec.handle(H)
        cfg = McCfg()
        cfg.frames, cfg.first_frame, cfg.snr, cfg.seed, cfg.noise = F, 0, 2.0, 1, 0
        cfg.codewords, cfg.n_codewords = cws.ctypes.data, 1
        check(lib().acg_ldpc_awgn_dev(h, C.byref(cfg), y.data_ptr(), stream.cuda_stream))
        torch.cuda.synchronize()
        ms = []
        for k in range(launches + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dec.decode_batch_dev(H, y.data_ptr(), False, F, 2.0, bits.data_ptr(), ok.data_ptr(), its.data_ptr(), stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            if k:
                ms.append(e0.elapsed_time(e1))
        probe = re.search(r"workspace_probe_ms=(\S+)", desc)
        print("%-28s base=%#x  mod2MiB=%#8x mod1GiB=%#10x  ms=%s  ok=%d%s" % (tag, base, base % (2 << 20), base % (1 << 30),
                                                                             " ".join("%.1f" % x for x in ms), int(ok.sum()),
                                                                             ("  probes of the candidates (ms): " + probe.group(1)) if probe else ""), flush=True)
        dec.close()
        del y, bits, ok, its
        return ms

    def churn(gb):
        t = torch.empty(gb << 30, dtype=torch.uint8, device="cuda")
        t.fill_(1)
        torch.cuda.synchronize()
        del t
        torch.cuda.empty_cache()

    if len(sys.argv) > 1 and sys.argv[1] == "alloc":
        # does a physically contiguous workspace (hipDeviceMallocContiguous) take the lottery out?
        for mode in (0, 1, 0, 1):
            os.environ["ACG_STREAM_WS_ALLOC"] = str(mode)
            for k in range(3):
                churn(48 if k % 2 == 0 else 16)
                run("alloc_mode=%d after_churn_%d" % (mode, k), 3)
                run("alloc_mode=%d recreate_%d" % (mode, k), 3)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "final":
        # the library default (chunks spaced out 16-fold, one candidate) against plain chunks, alternating over allocation histories
        for rep in range(4):
            for sp in ("1", None):
                if sp is None:
                    os.environ.pop("ACG_STREAM_WS_SPREAD", None)
                else:
                    os.environ["ACG_STREAM_WS_SPREAD"] = sp
                if rep:
                    churn(16 if rep % 2 else 48)
                t0 = time.time()
                run("%s rep %d" % ("plain chunks" if sp else "default (spread 16)", rep), 3)
                print("   (decoder creation + 4 launches: %.1f s)" % (time.time() - t0), flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "spread":
        # is the slow mode a COMPACT physical region (few DRAM banks / rows in play)?  keep every K-th of K times as many chunks
        os.environ["ACG_STREAM_WS_TRIES"] = "3"
        for rep in range(2):
            for k in (1, 4, 16, 30):
                os.environ["ACG_STREAM_WS_SPREAD"] = str(k)
                t0 = time.time()
                run("spread=%d rep %d" % (k, rep), 2)
                print("   (%.1f s)" % (time.time() - t0), flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "tries":
        # best-of-K workspace candidates (what the library does by default, K = 3) against K = 1, alternating
        for rep in range(4):
            for k in (1, 3, 5):
                os.environ["ACG_STREAM_WS_TRIES"] = str(k)
                churn(16 if rep % 2 else 48)
                run("tries=%d rep %d" % (k, rep), 3)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "contig_pads":
        # a physically contiguous workspace has a DETERMINISTIC layout: is there a slab stride that makes it fast?
        os.environ["ACG_STREAM_WS_ALLOC"] = "1"
        base = 10400000
        for pad in (0, 256, 1024, 2048, 3840, 4096 + 3840, 16384 - base % 16384, 65536 - base % 65536, 65536 - base % 65536 + 256,
                    (1 << 20) - base % (1 << 20), (1 << 20) - base % (1 << 20) + 4096, (2 << 20) - base % (2 << 20), (2 << 20) - base % (2 << 20) + 256,
                    (2 << 20) - base % (2 << 20) + 65536, 777 * 256, 12345 * 256):
            os.environ["ACG_STREAM_SLAB_PAD"] = str(pad)
            run("contiguous stride=%d (+%d)" % (base + pad, pad), 2)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "chunks":
        # workspace made of separately created physical chunks mapped in a shuffled order (mode 2) / in creation order (mode 3)
        for mode, mb in ((2, 2), (2, 32), (2, 256), (3, 32), (1, 0), (2, 32), (0, 0)):
            os.environ["ACG_STREAM_WS_ALLOC"] = str(mode)
            os.environ["ACG_STREAM_WS_CHUNK_MB"] = str(mb)
            for k in range(2):
                churn(48 if k % 2 == 0 else 16)
                t0 = time.time()
                run("mode=%d chunk=%dMB after_churn_%d" % (mode, mb, k), 3)
                run("mode=%d chunk=%dMB recreate_%d (%.1fs)" % (mode, mb, k, time.time() - t0), 3)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "pads":
        # slab stride sensitivity: for every pad, the decoder created right after a 48 GB allocate / free (the history that gave
        # the slow mode) and created again straight away (the history that gave the fast one)
        for pad in (0, 256, 3840, 4096 + 3840, 65536 + 3840, 2 * 1048576 - 10400000 % (2 * 1048576), 1 << 20):
            os.environ["ACG_STREAM_SLAB_PAD"] = str(pad)
            churn(48)
            a = run("pad=%d after_free_48GB" % pad, 3)
            b = run("pad=%d recreate" % pad, 3)
            churn(16)
            c = run("pad=%d after_free_16GB" % pad, 3)
        return
    run("fresh")
    run("recreate_1")
    t = torch.empty(48 << 30, dtype=torch.uint8, device="cuda")
    t.fill_(1)
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()
    run("after_free_48GB")
    hold = torch.empty(64 << 30, dtype=torch.uint8, device="cuda")
    hold.fill_(1)
    run("while_holding_64GB")
    hold2 = torch.empty(64 << 30, dtype=torch.uint8, device="cuda")
    hold2.fill_(2)
    run("while_holding_128GB")
    del hold, hold2
    torch.cuda.empty_cache()
    run("after_release")
    # many small allocations first (fragmented free list), then the workspace
    small = [torch.empty(3 << 20, dtype=torch.uint8, device="cuda") for _ in range(2000)]
    del small[::2]
    run("after_fragmenting")
    del small
    torch.cuda.empty_cache()
    run("recreate_last")


if __name__ == "__main__":
    main()
