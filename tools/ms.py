"""prints ms_per_step of the bench JSON line on stdin (in-step A/B runs: tools/ms.py < bench output)"""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print(f"{d['ms_per_step']:.4f} ms  {d['value']:.0f} pairs/s  roofline {d['roofline']['frac']:.3f}  loss {d.get('final_loss')}")
