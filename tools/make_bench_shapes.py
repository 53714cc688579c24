#!/usr/bin/env python
"""Writes tests/golden/bench_shapes.json: every DISTINCT launch of the timed training step at the bench's own sizes.

Runs one eager step of the AtomNAS-C supernet (the headline workload, BASELINE.json config 4) and of the AtomNAS-A supernet (config 2)
at batch 256 / 224 x 224 / bf16 with atomnas_amd.ops.RECORD on: every wrapper of the C ABI then appends a dict that names its entry point
and everything that selects a kernel instance or a launch geometry (sizes, prologue / epilogue / statistics modes, activation layouts and
pitches, workspace sizes).  Rows are de-duplicated; `nets` says which network launches the row, `count` how often per step.
tests/test_bench_shapes_gpu.py replays every row on seeded random data against torch on the GPU.

    python tools/make_bench_shapes.py [out.json]        (GPU box; ~1 minute)
"""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from atomnas_amd import ops  # noqa: E402


def record(model_name, batch):
    model, ts, hp, opt, ema, pinfo = bench.build(model_name, torch.bfloat16, batch, 1995)
    ts.use_graph = False
    g = torch.Generator(device="cuda").manual_seed(1995)
    ts.set_batch(torch.randn(batch, 3, hp['image_size'], hp['image_size'], device="cuda", generator=g),
                 torch.randint(0, 1000, (batch,), device="cuda", generator=g))
    ts.step(rho=1e-5)          # first step: allocations, plan scratch
    torch.cuda.synchronize()
    ops.RECORD = []
    ts.step(rho=1e-5)
    torch.cuda.synchronize()
    rows, ops.RECORD = ops.RECORD, None
    del model, ts
    torch.cuda.empty_cache()
    return rows


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "bench_shapes.json")
    batch = int(os.environ.get("BENCH_SHAPES_BATCH", "256"))
    table = collections.OrderedDict()
    for net in ("atomnas_c_supernet", "atomnas_a_supernet"):
        for r in record(net, batch):
            key = json.dumps(r, sort_keys=True)
            e = table.setdefault(key, dict(row=r, nets=collections.OrderedDict()))
            e["nets"][net] = e["nets"].get(net, 0) + 1
    rows = []
    for e in table.values():
        r = dict(e["row"])
        r["nets"] = e["nets"]
        rows.append(r)
    with open(out, "w") as f:
        f.write("[\n" + ",\n".join(json.dumps(r, sort_keys=True) for r in rows) + "\n]\n")
    per = collections.Counter(r["entry"] for r in rows)
    print("%d distinct launches -> %s" % (len(rows), out))
    for k, v in sorted(per.items()):
        print("  %-20s %d" % (k, v))


if __name__ == "__main__":
    main()
