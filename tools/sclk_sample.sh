#!/bin/bash
# Samples the GPU's current shader-clock level (sysfs pp_dpm_sclk, the line marked *) while a command runs.
#   tools/sclk_sample.sh <out file> <command...>
out="$1"; shift
f=$(ls /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -1)
"$@" &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  if [ -n "$f" ]; then grep '\*' "$f" | tr '\n' ' ' >> "$out"; echo >> "$out"; fi
  sleep 0.2
done
wait $pid
