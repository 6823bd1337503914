#!/usr/bin/env python3
"""Where does the host-fed training loop lose its time?  (value_with_input_transfer was 30 % below value
although 12.6 MB per step is far below PCIe.)  Measures, on one GPU:
  1. pinned H2D copy bandwidth at the sizes involved (one copy at a time, HIP events);
  2. the C2 training loop: resident batches | two fp32 H2D copies + stage kernel (round 3) | ONE packed copy (bf16 EEG
     operand + fp32 volumes, 8.4 MB) + one D2D | the copies alone in the same event structure | the packed loop with the
     H2D issued but never waited for (diagnostic: what the cross-stream dependency itself costs).
usage: python tools/h2d_probe.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bandwidth():
    print("== pinned host -> device, one copy at a time")
    for mb in (1.0, 4.2, 8.4, 12.6, 33.6, 134.0):
        n = int(mb * 1e6)
        h = torch.empty(n, dtype=torch.uint8).pin_memory()
        d = torch.empty(n, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            d.copy_(h, non_blocking=True)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print(f"  {mb:7.1f} MB: {ms * 1e3:8.1f} us  {n / ms / 1e6:7.2f} GB/s")


def loops(steps):
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    torch.manual_seed(0)
    tr = BridgeTrainer(eeg_channels=64, dropout=0.3).train()
    NB = 4
    dev = [synthetic_pairs(32, 64, 1024, (32, 32, 32), seed=1234 + 1000 * i) for i in range(NB)]
    tr.train_step(*dev[0])
    for i in range(300):
        tr.train_step(*dev[i % NB])
    torch.cuda.synchronize()
    import gc
    gc.collect(); gc.freeze(); gc.disable()

    def timed(name, body, warm=20):
        body(warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        body(steps)
        t_host = time.perf_counter() - t0                      # the host has ISSUED everything (it runs ahead of the GPU)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"  {name:78s} {dt * 1e3:7.4f} ms/step  {32 / dt:9.0f} pairs/s   (host issue {t_host / steps * 1e3:6.3f} ms/step)", flush=True)
        return dt

    print(f"== C2 training loop, {steps} steps each")
    base = timed("resident batches (mm_stage_inputs + replay)", lambda n: [tr.train_step(*dev[i % NB]) for i in range(n)])
    # the same steps with the host given a head start: K steps are enqueued while the GPU spins, then timed by device
    # events - the step's pure GPU time, with every packet written before it is needed
    if os.environ.get("MM_PROBE_HEADSTART", "1") == "1":
        for K, spin_ms in ((40, 40),):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            torch.cuda._sleep(int(spin_ms * 2.1e6))            # ~spin_ms at ~2.1 GHz
            e0.record()
            for i in range(K):
                tr.train_step(*dev[i % NB])
            e1.record()
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            print(f"  {K} steps enqueued behind a {spin_ms} ms spin kernel: {e0.elapsed_time(e1) / K:7.4f} ms/step by device events "
                  f"(host needed {t_issue * 1e3:.2f} ms to enqueue them, everything done after {t_all * 1e3:.2f} ms)", flush=True)
    host = [(e.cpu().pin_memory(), f.cpu().pin_memory()) for e, f in dev]
    packed = [tr.pack_host_batch(e, f) for e, f in dev]
    copy_s = torch.cuda.Stream()
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    consumed = [torch.cuda.Event(), torch.cuda.Event()]
    stage2 = [(torch.empty_like(dev[0][0]), torch.empty_like(dev[0][1])) for _ in range(2)]
    stage1 = [torch.empty(packed[0].numel(), dtype=torch.uint8, device="cuda") for _ in range(2)]
    for b in range(2):
        consumed[b].record()

    def loop_two(n, train=True):
        def up(i):
            b = i % 2
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(consumed[b])
                stage2[b][0].copy_(host[i % NB][0], non_blocking=True)
                stage2[b][1].copy_(host[i % NB][1], non_blocking=True)
                ready[b].record(copy_s)
        up(0)
        for i in range(n):
            if i + 1 < n:
                up(i + 1)
            torch.cuda.current_stream().wait_event(ready[i % 2])
            if train:
                tr.train_step(*stage2[i % 2])
            consumed[i % 2].record()

    def loop_one(n, train=True, wait=True):
        def up(i):
            b = i % 2
            with torch.cuda.stream(copy_s):
                if wait:
                    copy_s.wait_event(consumed[b])
                stage1[b].copy_(packed[i % NB], non_blocking=True)
                ready[b].record(copy_s)
        up(0)
        for i in range(n):
            if i + 1 < n:
                up(i + 1)
            if wait:
                torch.cuda.current_stream().wait_event(ready[i % 2])
            if train:
                tr.train_step_packed(stage1[i % 2])
            consumed[i % 2].record()

    def loop_hostsync(n, depth=3, record_main=True):
        """the same copies ordered from the HOST: the loop blocks on `ready` (an event on the copy stream, recorded a whole step
        ago) before it launches the step, and on `consumed` (recorded on the step's stream two steps ago) before it reuses a
        staging buffer - no hipStreamWaitEvent anywhere, so neither queue ever holds a barrier for the other.  The host has
        the time (0.25 ms of work per 0.77 ms step)."""
        ring = [torch.empty(packed[0].numel(), dtype=torch.uint8, device="cuda") for _ in range(depth)]
        rdy = [torch.cuda.Event() for _ in range(depth)]
        con = [torch.cuda.Event() for _ in range(depth)]
        for e in con:
            e.record()

        def up(i):
            b = i % depth
            if record_main:
                con[b].synchronize()
            with torch.cuda.stream(copy_s):
                ring[b].copy_(packed[i % NB], non_blocking=True)
                rdy[b].record(copy_s)
        up(0)
        for i in range(n):
            if i + 1 < n:
                up(i + 1)
            rdy[i % depth].synchronize()
            tr.train_step_packed(ring[i % depth])
            if record_main:
                con[i % depth].record()

    def loop_deep(n, depth=4):
        """packed copies into a ring of `depth` staging buffers, issued depth - 1 steps ahead: the buffer-reuse guard is then
        always an already-completed event (no real cross-queue dependency on the copy stream)"""
        ring = [torch.empty(packed[0].numel(), dtype=torch.uint8, device="cuda") for _ in range(depth)]
        rdy = [torch.cuda.Event() for _ in range(depth)]
        con = [torch.cuda.Event() for _ in range(depth)]
        for e in con:
            e.record()

        def up(i):
            b = i % depth
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(con[b])
                ring[b].copy_(packed[i % NB], non_blocking=True)
                rdy[b].record(copy_s)
        for i in range(min(depth - 1, n)):
            up(i)
        for i in range(n):
            if i + depth - 1 < n:
                up(i + depth - 1)
            torch.cuda.current_stream().wait_event(rdy[i % depth])
            tr.train_step_packed(ring[i % depth])
            con[i % depth].record()

    # the same packed loop with events that release to DEVICE scope only (hipEventReleaseToDevice; torch.cuda.Event has no
    # such flag: raw HIP events through ctypes).  A default event record is a system-scope release (cache write-back).
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hipEventDisableTiming, hipEventReleaseToDevice = 0x2, 0x40000000

    class RawEvent:
        def __init__(self, flags):
            self.h = ctypes.c_void_p()
            assert hip.hipEventCreateWithFlags(ctypes.byref(self.h), ctypes.c_uint(flags)) == 0

        def record(self, stream):
            assert hip.hipEventRecord(self.h, ctypes.c_void_p(stream.cuda_stream)) == 0

        def wait(self, stream):
            assert hip.hipStreamWaitEvent(ctypes.c_void_p(stream.cuda_stream), self.h, ctypes.c_uint(0)) == 0

    def loop_raw(n, flags):
        rdy = [RawEvent(flags) for _ in range(2)]
        con = [RawEvent(flags) for _ in range(2)]
        main = torch.cuda.current_stream()
        for e in con:
            e.record(main)

        def up(i):
            b = i % 2
            con[b].wait(copy_s)
            with torch.cuda.stream(copy_s):
                stage1[b].copy_(packed[i % NB], non_blocking=True)
            rdy[b].record(copy_s)
        up(0)
        for i in range(n):
            if i + 1 < n:
                up(i + 1)
            rdy[i % 2].wait(main)
            tr.train_step_packed(stage1[i % 2])
            con[i % 2].record(main)

    timed("round 3: two fp32 H2D copies (12.6 MB) + stage kernel + replay", loop_two)
    timed("   the same event / copy structure without the training step", lambda n: loop_two(n, False))
    timed("ONE packed H2D copy (bf16 EEG operand + fp32 volumes, 8.4 MB) + one D2D + replay", loop_one)
    timed("   the same event / copy structure without the training step", lambda n: loop_one(n, False))
    timed("   packed copy issued but never waited for (diagnostic: races on purpose)", lambda n: loop_one(n, True, False))
    timed("resident packed buffers (train_step_packed on a device copy: D2D + replay)",
          lambda n: [tr.train_step_packed(stage1[i % 2]) for i in range(n)])
    timed("ONE packed copy, ring of 4 staging buffers, copies issued 3 steps ahead", loop_deep)
    timed("ONE packed copy, ring of 3 staging buffers, copies issued 2 steps ahead", lambda n: loop_deep(n, 3))
    timed("ONE packed copy, raw HIP events, default flags (disable timing)", lambda n: loop_raw(n, hipEventDisableTiming))
    timed("ONE packed copy, raw HIP events with hipEventReleaseToDevice", lambda n: loop_raw(n, hipEventDisableTiming | hipEventReleaseToDevice))
    timed("ONE packed copy, ordered by HOST-side event waits only (3 staging buffers)", lambda n: loop_hostsync(n))
    timed("   the same without the per-step event record on the step's stream (reuse guard dropped: diagnostic)", lambda n: loop_hostsync(n, record_main=False))
    timed("resident batches again", lambda n: [tr.train_step(*dev[i % NB]) for i in range(n)])
    # same packed loop, H2D on the MAIN stream (no second stream, no events): copy and step serialised
    def serial(n):
        for i in range(n):
            stage1[0].copy_(packed[i % NB], non_blocking=True)
            tr.train_step_packed(stage1[0])
    timed("packed copy on the step's own stream (serialised, no events)", serial)
    return base


if __name__ == "__main__":
    bandwidth()
    loops(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
