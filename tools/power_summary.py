"""exploratory: mean / max of the rocm-smi samples tools/power_probe.sh wrote:  tools/power_summary.py file ..."""
import json, sys
for f in sys.argv[1:]:
    pw, ck = [], []
    for line in open(f):
        line = line.strip()
        if not line.startswith("{"): continue
        try: d = json.loads(line)["card0"]
        except Exception: continue
        pw.append(float(d["Current Socket Graphics Package Power (W)"])); ck.append(int(d["sclk clock speed:"].strip("()Mhz")))
    busy = [(p, c) for p, c in zip(pw, ck) if p > 600]
    if busy: print(f, "samples", len(busy), "power mean %.0f max %.0f W" % (sum(p for p, _ in busy) / len(busy), max(p for p, _ in busy)), "sclk mean %.0f MHz" % (sum(c for _, c in busy) / len(busy)))
    else: print(f, "no busy samples", len(pw))
