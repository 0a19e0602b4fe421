#!/bin/bash
# 256 MiB of 8 KiB packets, 64 stream ids round robin (1/64 of them filtered): what a block that
# filters ONE id of a shared pipe mostly does is forward.  Ours vs the reference's hilbert.
set -u
python3 - <<'PY'
import struct, numpy as np
pay = np.sin(np.arange(2044, dtype=np.float32) * 0.3).astype(np.float32).tobytes()
with open('/dev/shm/pkts64.bin', 'wb') as f:
    for i in range((256 << 20) // (16 + len(pay))):
        f.write(struct.pack('<HHIQ', i % 64, 6, len(pay), 2044000) + pay)
PY
for b in oracle/_ref/hilbert blocks/bin/hilbert; do
  for i in 1 2; do
    s=$(date +%s%N); cat /dev/shm/pkts64.bin | timeout -k 5 120 $b 0 | cat > /dev/null; e=$(date +%s%N)
    echo "$b: $(( (e - s) / 1000000 )) ms for 256 MiB (63/64 forwarded)"
  done
done
cat /dev/shm/pkts64.bin | oracle/_ref/hilbert 0 | wc -c
cat /dev/shm/pkts64.bin | blocks/bin/hilbert 0 | wc -c
rm -f /dev/shm/pkts64.bin
