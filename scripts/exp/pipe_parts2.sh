#!/bin/bash
N=$((131329 * 8192))
t() { local s=$(date +%s%N); eval "$1"; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms : $1"; }
G="blocks/bin/gen freq=100k waveform=sine sample_rate=1M mode=real"
for gb in 128 512 2048 8192; do for hb in 2097152 8388608 33554432; do
  export SDR_AMD_GEN_BLOCKS=$gb SDR_AMD_BATCH_BYTES=$hb
  t "$G | head -c $N | blocks/bin/hilbert > /dev/null  # gen blocks $gb, hilbert batch $hb"
done; done
