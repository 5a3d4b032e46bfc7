#!/bin/bash
# Round 3: one-variable sweeps of the solo-latency switches (tools/trace_rounds.py prints the per-round waits).
# Usage: bash tools/r3_sweep.sh <tag> <bn> "VAR=a VAR=b ..."   (each entry is an env assignment list joined by commas)
TAG=$1; BN=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in "$@"; do
  envs=$(echo $v | tr ',' ' ')
  f=$OUT/bn${BN}_$(echo $v | tr ',=' '__')
  env $envs timeout 300 python tools/trace_rounds.py $BN 3 2> $f.txt
  echo "== $v: $(grep prove $f.txt | tr '\n' ' ')"
  tail -n +$(( $(grep -n "rounds trace" $f.txt | tail -1 | cut -d: -f1) )) $f.txt | grep -v prove | head -24
done
