import sys; sys.path.insert(0, "probes"); import pbench
for b in (1, 4, 8, 16, 32, 48, 64, 96, 128, 160, 192, 256): pbench.run(b, "bf16", reps=3 if b > 128 else 5)
for b in (16, 64, 128): pbench.run(b, "fp32", reps=3)
