"""Call sites of torch.cat / torch.stack / zeros / zeros_like / new_zeros / full in one train step of the default workload (a Python-side count: which lines
of this repo ask for the small copy and fill kernels).  Run on the GPU box: python tools/debug/cat_census.py"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    import bench
    sys.argv = ["bench.py"]
    args = bench.parse()
    from monosowa_amd import miopen_tuning
    miopen_tuning.use_shipped_db(0)
    import torch
    device = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = False
    from monosowa_amd.synthetic import make_batch
    cfg, model, criterion, optimizer, (W, H) = bench.build_everything(args, device)
    model.train(True)
    criterion.train(True)
    batch = make_batch(args.batch, device, seed=444, resolution=(W, H), mixed_cameras=False)
    batch = (batch[0].contiguous(memory_format=torch.channels_last),) + batch[1:]
    step = bench.train_step_fn(model, criterion, optimizer)
    for _ in range(2):
        step(batch)
    torch.cuda.synchronize()
    counts = collections.Counter()
    active = [False]

    def site():
        for fr in reversed(traceback.extract_stack()[:-2]):
            if "monosowa_amd" in fr.filename or fr.filename.endswith("bench.py"):
                return "%s:%d" % (fr.filename.split("monosowa_amd/")[-1], fr.lineno)
        return "?"

    def wrap(mod, name):
        orig = getattr(mod, name)

        def f(*a, **k):
            out = orig(*a, **k)
            if active[0] and isinstance(out, torch.Tensor) and out.is_cuda:
                counts[(name, site(), tuple(out.shape))] += 1
            return out
        setattr(mod, name, f)
    for n in ("cat", "stack", "zeros", "zeros_like", "full", "full_like", "ones", "ones_like", "empty_like", "arange", "tensor", "as_tensor"):
        wrap(torch, n)
    active[0] = True
    step(batch)
    torch.cuda.synchronize()
    active[0] = False
    for (name, where, shape), n in sorted(counts.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print("%3d  %-10s %-50s %s" % (n, name, where, list(shape)))


if __name__ == "__main__":
    main()
