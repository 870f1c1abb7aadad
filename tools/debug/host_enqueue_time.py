"""Host time to ENQUEUE one train step of the default workload (no synchronisation inside the loop; the first steps after a
device synchronisation, before any queue limit can throttle the host) against the step's GPU time.
    python tools/debug/host_enqueue_time.py"""
import os
import sys
import time

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
    for _ in range(5):
        step(batch)
    torch.cuda.synchronize()
    for trial in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks = []
        for _ in range(4):
            step(batch)
            marks.append(time.perf_counter())
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        host = [1e3 * (b - a) for a, b in zip([t0] + marks[:-1], marks)]
        print("trial %d: host ms per step() call %s; 4 steps on the device %.1f ms (%.1f per step)"
              % (trial, ", ".join("%.1f" % h for h in host), 1e3 * (t_end - t0), 1e3 * (t_end - t0) / 4))


if __name__ == "__main__":
    main()
