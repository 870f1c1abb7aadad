"""What the encoder's sampling locations look like in the train step bench.py times, and what the cell scatter's scan makes of them.

    python tools/debug/offset_stats.py [--steps 30] [--batch 16]

Runs bench.py's train step for --steps steps, keeps the locations the fused forward saved for the backward (level-major
[B, M, L, Lq, P, 2]) of every encoder layer in the last step, and feeds batch element 0 to the scan model of
tools/debug/scan_sim.py: candidates scanned / delivering, point tests / delivered, with the plan's per-(head, level) bounds and with
per-(head, level, point) bounds; plus the spread of d per (head, level, point)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "debug"))
import bench  # noqa: E402
import scan_sim  # noqa: E402
from monosowa_amd import MultiScaleDeformableAttention as MSDA  # noqa: E402
from monosowa_amd import miopen_tuning  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--resolution", default="1280x384")
    a = ap.parse_args()
    miopen_tuning.use_shipped_db(0)
    sys.argv = [sys.argv[0], "--batch", str(a.batch), "--resolution", a.resolution]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    cfg, model, criterion, optimizer, (W, H) = bench.build_everything(args, dev)
    model.train(); criterion.train()
    from monosowa_amd.synthetic import make_batch
    batch = make_batch(a.batch, dev, seed=444, resolution=(W, H))
    batch = (batch[0].contiguous(memory_format=torch.channels_last),) + batch[1:]
    step = bench.train_step_fn(model, criterion, optimizer)
    captured = []
    orig = MSDA.ms_deform_attn_fused_forward_merged_save

    def spy(v, shapes, lsi, proj, ref):
        out = orig(v, shapes, lsi, proj, ref)
        captured.append((out[1][0].detach().cpu().numpy(), shapes.cpu().numpy()))       # batch element 0: [M, L, Lq, P, 2]
        return out
    for i in range(a.steps):
        if i == a.steps - 1:
            MSDA.ms_deform_attn_fused_forward_merged_save = spy
            import monosowa_amd.encoder_block as EB
            EB.MSDA.ms_deform_attn_fused_forward_merged_save = spy
        step(batch)
    torch.cuda.synchronize()
    MSDA.ms_deform_attn_fused_forward_merged_save = orig
    print("captured %d encoder layers after %d train steps" % (len(captured), a.steps))
    for li, (loc, shapes) in enumerate(captured):
        Hs, Ws = [int(x) for x in shapes[:, 0]], [int(x) for x in shapes[:, 1]]
        starts = np.concatenate([[0], np.cumsum([h * w for h, w in zip(Hs, Ws)])])
        tok = np.arange(starts[-1])
        lq_of = np.searchsorted(starts[1:], tok, side="right")
        rel = tok - starts[lq_of]
        yq_of = rel // np.array(Ws)[lq_of]; xq_of = rel - yq_of * np.array(Ws)[lq_of]

        def cells_of(m, l, loc=loc, Hs=Hs, Ws=Ws):
            x = loc[m, l, :, :, 0].astype(np.float32); y = loc[m, l, :, :, 1].astype(np.float32)      # [Lq, P]
            wim = x * np.float32(Ws[l]) - np.float32(0.5); him = y * np.float32(Hs[l]) - np.float32(0.5)
            ok = (him > -1) & (wim > -1) & (him < Hs[l]) & (wim < Ws[l])
            q, p = np.nonzero(ok)
            return lq_of[q], yq_of[q], xq_of[q], p, np.floor(him[ok]).astype(int), np.floor(wim[ok]).astype(int)
        print("=== encoder layer %d" % li)
        # spread of the offsets around the query's own centre, in pixels of the sampled level
        for l in range(4):
            row = []
            for m in range(8):
                x = loc[m, l, :, :, 0] * Ws[l] - 0.5; y = loc[m, l, :, :, 1] * Hs[l] - 0.5
                cx = (xq_of + 0.5) / np.array(Ws)[lq_of] * Ws[l] - 0.5; cy = (yq_of + 0.5) / np.array(Hs)[lq_of] * Hs[l] - 0.5
                dx = x - cx[:, None]; dy = y - cy[:, None]
                row.append("m%d x %s y %s" % (m, " ".join("%+.1f~%.2f" % (dx[:, p].mean(), dx[:, p].std()) for p in range(4)),
                                              " ".join("%+.1f~%.2f" % (dy[:, p].mean(), dy[:, p].std()) for p in range(4))))
            print("level %d offsets mean~std per point:\n   " % l + "\n   ".join(row[:2]))
        scan_sim.analyse(Hs, Ws, cells_of)


if __name__ == "__main__":
    main()
