#!/bin/bash
# samples the GPU's clocks / power from sysfs while tools/kb_base runs a long sustained sequence of 8K launches
D=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -1)
H=$(ls -d $D/hwmon/hwmon* 2>/dev/null | head -1)
echo "device dir $D hwmon $H"
ls $D | grep -i "pp_dpm\|power\|gpu_busy\|mem_busy" | tr '\n' ' '; echo
for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk; do [ -r $D/$f ] && { echo "== $f (idle)"; cat $D/$f; }; done
[ -n "$H" ] && ls $H | tr '\n' ' '; echo
KB_N=${KB_N:-2500} KB_REPS=2 KB_NSET=3 ./tools/kb_base ${1:-C} gauss > /tmp/kb.out 2>&1 &
pid=$!
for i in $(seq 1 ${SAMPLES:-40}); do
  s=$(grep '\*' $D/pp_dpm_sclk 2>/dev/null | tr -d '\n'); m=$(grep '\*' $D/pp_dpm_mclk 2>/dev/null | tr -d '\n'); f=$(grep '\*' $D/pp_dpm_fclk 2>/dev/null | tr -d '\n')
  p=$(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null); fr=$(cat $H/freq1_input 2>/dev/null)
  echo "t=$i sclk[$s] mclk[$m] fclk[$f] power_uW=$p freq1=$fr"
  sleep 0.02
  kill -0 $pid 2>/dev/null || break
done
wait $pid
cat /tmp/kb.out
