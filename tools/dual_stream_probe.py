#!/usr/bin/env python3
"""Probe: does the chip finish 12 UNet samples sooner as TWO captured batch-6 evaluations replayed concurrently on two streams than as ONE
batch-12 evaluation?  (Timing only: lanes share shape-keyed scratch, results of the concurrent arm are not checked here.)
    python tools/dual_stream_probe.py [reps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from anyedit_amd.anysd.pipeline import EditPipeline  # noqa: E402


def make_pipe(moe, sched, B, dev, rank):
    x_T, img_lat, ehs, null, ref, code = bench.synthetic_inputs(B, dev, rank)
    p = EditPipeline(moe, sched, use_graph=True)
    p.prepare(img_lat, ehs, null, ref, code)
    p._x_in[:, :4].copy_(torch.cat([x_T] * 3, 0))
    p.set_step(500)
    p._ensure_graph(("probe", B))
    return p


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    unet, moe, sched = bench.build_model(dev)
    with torch.no_grad():
        p12 = make_pipe(moe, sched, 4, dev, 0)
        pa = make_pipe(moe, sched, 2, dev, 1)
        pb = make_pipe(moe, sched, 2, dev, 2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def one12():
        p12._graph.replay()

    def seq66():
        pa._graph.replay()
        pb._graph.replay()

    def par66():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            pa._graph.replay()
        with torch.cuda.stream(s2):
            pb._graph.replay()
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    for name, fn in (("one batch-12 graph", one12), ("two batch-6 graphs, one stream", seq66), ("two batch-6 graphs, two streams", par66),
                     ("one batch-12 graph", one12), ("two batch-6 graphs, two streams", par66)):
        print(f"{name:36s} {timed(fn):8.3f} ms per 12 samples", flush=True)


if __name__ == "__main__":
    main()
