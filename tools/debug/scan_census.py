"""The cell scatter's scan census in the train step bench.py times: candidates scanned / delivering, point tests / delivered, points
skipped by the per-point reach test -- per encoder-shape backward launch.  Needs a measurement build of the library:

    tools/debug/build_variant.sh count -DMSDA_ROWS_COUNT=1
    MONOSOWA_MSDA_LIB=tools/debug/variants/count.so python tools/debug/scan_census.py [--steps 30] [--micro init|normal:2|...]

--micro: the operator pair of tools/msda_fused_bench.py on synthetic offsets instead of the train step."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from monosowa_amd import _lib  # noqa: E402
from monosowa_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402

NAMES = ("scan_candidates", "scan_delivering", "scan_point_tests", "scan_delivered", "scan_points_skipped")


def report(tag, launches):
    c = {n: _lib.debug_counter(n) for n in NAMES}
    if not c["scan_candidates"]:
        print("%s: no counts -- is MONOSOWA_MSDA_LIB a -DMSDA_ROWS_COUNT=1 build?" % tag)
        return
    per = {n: v / launches for n, v in c.items()}
    print("%s: per launch (%d launches): candidates %.2f M, delivering %.2f M (%.0f %%); point tests %.2f M, delivered %.2f M (%.0f %% of "
          "tests, %.0f %% of candidates x 4); skipped before their loads %.2f M points"
          % (tag, launches, per["scan_candidates"] / 1e6, per["scan_delivering"] / 1e6, 100 * per["scan_delivering"] / per["scan_candidates"],
             per["scan_point_tests"] / 1e6, per["scan_delivered"] / 1e6, 100 * per["scan_delivered"] / max(per["scan_point_tests"], 1),
             100 * per["scan_delivered"] / (4 * per["scan_candidates"]), per["scan_points_skipped"] / 1e6), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--micro", default=None)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--resolution", default=None, help="--micro only: WxH (config 5: 1920x1280 with --batch 4)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    if a.micro:
        import msda_fused_bench as FB
        if a.resolution:
            FB.set_resolution(a.resolution)
        for spec in a.micro.split(","):
            value, shapes, lsi, proj, ref, go = FB.make(a.batch, "enc", spec, dev)
            _, loc, attw = MSDA.ms_deform_attn_fused_forward_merged_save(value, shapes, lsi, proj, ref)
            torch.cuda.synchronize()
            for n in NAMES:
                _lib.debug_counter(n)
            for _ in range(4):
                MSDA.ms_deform_attn_fused_backward_merged_saved(value, shapes, lsi, loc, attw, ref, go)
            report("micro %s" % spec, 4)
        return
    import bench
    from monosowa_amd import miopen_tuning
    miopen_tuning.use_shipped_db(0)
    sys.argv = [sys.argv[0], "--batch", str(a.batch)]
    args = bench.parse()
    cfg, model, criterion, optimizer, (W, H) = bench.build_everything(args, dev)
    model.train(); criterion.train()
    from monosowa_amd.synthetic import make_batch
    batch = make_batch(a.batch, dev, seed=444, resolution=(W, H))
    batch = (batch[0].contiguous(memory_format=torch.channels_last),) + batch[1:]
    step = bench.train_step_fn(model, criterion, optimizer)
    for i in range(a.steps):
        step(batch)
    torch.cuda.synchronize()
    for n in NAMES:
        _lib.debug_counter(n)
    for i in range(3):
        step(batch)
    report("train step %d-%d" % (a.steps, a.steps + 2), 9)


if __name__ == "__main__":
    main()
