#!/bin/bash
# What shader clock does the part run while (a) the decode bench and (b) the issue-rate microbenchmark
# are running?  (Round-2 review item 2d / weak #8: every "floor" in DESIGN.md section 5 hangs on the
# time one wave64 vector instruction takes; settle whether that is cycles or clock.)
# Samples the driver's own view (sysfs pp_dpm_sclk / rocm-smi) a few times per second while the load runs.
#   bash tools/clock_sample.sh > gpurun_out/clocks.txt
R=$PWD
sample() {  # tag, seconds
  for i in $(seq $(( $2 * 4 ))); do
    for f in /sys/class/drm/card*/device/pp_dpm_sclk; do
      [ -r "$f" ] && echo "$1 sysfs $(grep '\*' $f | tr -d '\n')"
    done
    for f in /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input; do
      [ -r "$f" ] && echo "$1 hwmon_freq1 $(cat $f)"
    done
    sleep 0.25
  done
}
smi() { (rocm-smi --showclocks 2>/dev/null || /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null) | grep -i "sclk\|mclk\|fclk" | sed "s/^/$1 smi /"; }
echo "== idle"; sample idle 1; smi idle
echo "== bench loop (7b-w4-s0, graph replay)"
python bench.py --steps 3000 --repeats 3 --no-cpu-baseline --no-sub-records --no-roofline > /tmp/clk_bench.json 2>/dev/null &
pid=$!
sleep 6   # import + build of the workload
sample bench 3; smi bench
wait $pid
grep '^{' /tmp/clk_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench value', d['value'], 'ms_per_step', d['ms_per_step'])"
if [ -x build/exp/issue_rate ]; then
  echo "== issue_rate microbenchmark"
  build/exp/issue_rate > /tmp/issue_rate.txt 2>&1 &
  pid=$!
  sleep 1
  sample issue 4; smi issue
  wait $pid
  head -30 /tmp/issue_rate.txt
fi
