#!/bin/bash
# Power / shader clock of GPU 0 sampled from sysfs (hwmon) every ~20 ms while bench.py runs; summary to stdout.
#   bash tools/power_trace.sh [steps]        (run on the GPU box)
STEPS=${1:-1500}
R=$(cd "$(dirname "$0")/.." && pwd)
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
P=$(ls $H/power1_average $H/power1_input 2>/dev/null | head -1)
F=$(ls $H/freq1_input 2>/dev/null | head -1)
echo "# hwmon $H power file $P freq file $F cap $(cat $H/power1_cap 2>/dev/null)"
python $R/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-roofline > /tmp/pt_bench.log 2>&1 &
BP=$!
: > /tmp/pt.log
while kill -0 $BP 2>/dev/null; do
  echo "$(date +%s.%N) $(cat $P 2>/dev/null) $(cat $F 2>/dev/null)" >> /tmp/pt.log
  sleep 0.02
done
tail -1 /tmp/pt_bench.log | cut -c1-200
python - <<'PY'
import numpy as np
rows = [l.split() for l in open('/tmp/pt.log') if len(l.split()) == 3]
t = np.array([float(r[0]) for r in rows]); p = np.array([float(r[1]) for r in rows]) / 1e6; f = np.array([float(r[2]) for r in rows]) / 1e6
busy = f > 1000
print(f"samples {len(p)} over {t[-1] - t[0]:.1f} s, busy samples {int(busy.sum())} (sclk > 1 GHz)")
print("power histogram (W, all samples):", np.histogram(p, bins=[0, 300, 500, 700, 900, 1000, 1100, 1200, 1300, 1400, 1600])[0].tolist())
for name, v in (("power W", p[busy]), ("sclk MHz", f[busy])):
    print(name, "min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
PY
