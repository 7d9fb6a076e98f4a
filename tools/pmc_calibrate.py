"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS box for the two access patterns of the stepper (MI355X_MICROARCH.md, HBM section:
"calibrate on a known byte count in your own access pattern"):
  wide    every lane loads 16 consecutive bytes (k_obs: tile rows, mirrors)           1 GiB read once -> known bytes
  narrow  every lane loads ONE 2-byte word, lanes 1 KiB apart (k_step: a cell of a different env's [env][cell] grid per lane)
          2^20 distinct 64-byte lines touched once -> the counter value / 2^20 = bytes counted per narrow request
Run under rocprofv3 --pmc FETCH_SIZE (and again with WRITE_SIZE); tools/pmc_calibrate_read.py turns the CSV into factors."""
import torch

dev = torch.device("cuda", 0)
n = 1 << 30
x = torch.empty(n, dtype=torch.uint8, device=dev)
x.fill_(1)
torch.cuda.synchronize()
# footprints far beyond L2 + the 256 MiB Infinity Cache; every line is touched exactly once per kernel
wide = x.view(torch.int32)
y = torch.empty_like(wide)
y.copy_(wide)                        # wide streaming read + write of 1 GiB: "elementwise/copy" kernel
torch.cuda.synchronize()
narrow = x.view(torch.int16)[::512]  # one 2-byte word every 1 KiB: 2^20 lines
z = torch.empty(narrow.shape, dtype=torch.int16, device=dev)
z.copy_(narrow)                      # strided gather: 2 MiB of useful data out of 2^20 touched lines
torch.cuda.synchronize()
print("wide bytes", n, "narrow lines", narrow.numel())
