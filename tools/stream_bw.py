#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this GPU: torch copy / read-only reduction over buffers far larger than the
256 MB infinity cache.  Reference point for the HBM-bound stages (preprocess).  GPU analysis tool."""
import torch


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


def main():
    for mb in (512, 1024, 2048):
        n = mb * 1024 * 1024 // 4
        x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
        y = torch.empty_like(x)
        t = timeit(lambda: y.copy_(x))
        print(f"copy   {mb:5d} MB read + {mb:5d} MB write: {t * 1e3:8.1f} us  {2 * mb / 1024 / (t * 1e-3):7.2f} GiB/ms = {2 * mb * 1.048576 / t:7.1f} GB/s")
        t = timeit(lambda: x.sum())
        print(f"reduce {mb:5d} MB read                  : {t * 1e3:8.1f} us  {mb * 1.048576 / t:7.1f} GB/s")
        t = timeit(lambda: y.fill_(1.0))
        print(f"fill   {mb:5d} MB write                 : {t * 1e3:8.1f} us  {mb * 1.048576 / t:7.1f} GB/s")
        del x, y


if __name__ == "__main__":
    main()
