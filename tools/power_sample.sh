#!/bin/bash
# Samples board power and shader clock (sysfs hwmon of the first amdgpu device) every 50 ms while a command runs:  power_sample.sh <tag> <command...>
# Prints median / max of the samples taken while the command was running.  Evidence for the power wall: profiles/r05_power_clock.md.
TAG=$1; shift
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
PW=$(ls $H/power1_average $H/power1_input 2>/dev/null | head -1)
FQ=$(ls $H/freq1_input 2>/dev/null | head -1)
"$@" > /tmp/ps_$TAG.out 2>&1 &
PID=$!
: > /tmp/ps_$TAG.samples
while kill -0 $PID 2>/dev/null; do
  p=$(cat $PW 2>/dev/null); f=$(cat $FQ 2>/dev/null)
  echo "$p $f" >> /tmp/ps_$TAG.samples
  sleep 0.05
done
wait $PID
python3 - "$TAG" <<'PY'
import sys
tag=sys.argv[1]
rows=[l.split() for l in open(f'/tmp/ps_{tag}.samples') if len(l.split())==2]
rows=rows[len(rows)//4:]   # skip the start-up quarter
if not rows:
    print(f'{{"tag": "{tag}", "error": "no hwmon samples"}}'); sys.exit()
pw=sorted(float(r[0])/1e6 for r in rows); fq=sorted(float(r[1])/1e6 for r in rows)
print('{"tag": "%s", "samples": %d, "power_W_median": %.0f, "power_W_max": %.0f, "sclk_MHz_median": %.0f, "sclk_MHz_min": %.0f, "sclk_MHz_max": %.0f}' % (tag, len(rows), pw[len(pw)//2], pw[-1], fq[len(fq)//2], fq[0], fq[-1]))
PY
tail -1 /tmp/ps_$TAG.out | cut -c1-200
