"""Practical streaming bound on this GPU for the layer sizes of the net: time of a device-to-device copy (read N bytes +
write N bytes) at the activation sizes of a batch-64 step, warm (the source was just written, as a layer's input is).
Used to place the 1x1 / stem launches against what the memory system delivers rather than the 8 TB/s nameplate."""
import sys
import torch

def main():
    dev = torch.device("cuda:0")
    for mb in (11, 22, 44, 88, 176):
        n = mb * 1000 * 1000 // 2
        src = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
        dst = torch.empty_like(src)
        for _ in range(5):
            dst.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            dst.copy_(src)
            src, dst = dst, src          # the next copy reads what was just written
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / reps
        print(f"copy {mb:4d} MB -> {mb:4d} MB: {us:7.2f} us  = {2 * mb / us * 1e-3 * 1e3:7.1f} GB/s (read+write)", flush=True)

if __name__ == "__main__":
    sys.exit(main())
