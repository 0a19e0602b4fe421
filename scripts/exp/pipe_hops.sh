#!/bin/bash
t() { local s=$(date +%s%N); eval "$1"; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms : $1"; }
head -c 1075847168 /dev/zero > /dev/shm/z.bin
t "cat /dev/shm/z.bin > /dev/null"
t "cat /dev/shm/z.bin | cat > /dev/null"
t "cat /dev/shm/z.bin | cat | cat > /dev/null"
t "cat /dev/shm/z.bin | cat | cat | cat > /dev/null"
t "cat /dev/shm/z.bin | cat | cat | cat | cat > /dev/null"
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; taskset -p $$ | head -2
rm -f /dev/shm/z.bin
