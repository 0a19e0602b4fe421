#!/bin/bash
# where does the time of `gen | head | hilbert` go?  (2^28 real samples = 1.07 GB of packets)
N=$((131329 * 8192))
t() { local s=$(date +%s%N); eval "$1"; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms : $1"; }
G="blocks/bin/gen freq=100k waveform=sine sample_rate=1M mode=real"
t "$G | head -c $N > /dev/null"
t "$G | head -c $N | cat > /dev/null"
t "$G | head -c $N > /dev/shm/gen.bin"
t "blocks/bin/hilbert < /dev/shm/gen.bin > /dev/null"
t "cat /dev/shm/gen.bin | blocks/bin/hilbert > /dev/null"
t "cat /dev/shm/gen.bin | blocks/bin/hilbert | cat > /dev/null"
t "$G | head -c $N | blocks/bin/hilbert > /dev/null"
t "$G | blocks/bin/hilbert | head -c $N > /dev/null"
t "cat /dev/shm/gen.bin | head -c $N > /dev/null"
rm -f /dev/shm/gen.bin
