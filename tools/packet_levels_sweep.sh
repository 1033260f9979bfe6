# deeper levels of the ray tree through the packet kernels (RT_PACKET_LEVELS > 1).  Both level loops honour the setting (the loop that
# runs ahead of the host hands the packets of a launch whose count only the device knows out in order, groupStride 0); the sweep
# times each setting under both loops
set -e
mkdir -p gpurun_out/pl
out=gpurun_out/pl/times.txt
: > $out
for L in 1 2 8; do
  echo "== RT_PACKET_LEVELS=$L (level loop ahead of the host)" >> $out
  RT_PACKET_LEVELS=$L CONFIG=C1 C1_SCENES=abdeghil timeout -k 10 120 python3 tools/config_times.py >> $out 2>&1
  RT_PACKET_LEVELS=$L CONFIG=C3 timeout -k 10 120 python3 tools/config_times.py >> $out 2>&1
  echo "== RT_NO_LEVELS_AHEAD=1 RT_PACKET_LEVELS=$L" >> $out
  RT_NO_LEVELS_AHEAD=1 RT_PACKET_LEVELS=$L CONFIG=C1 C1_SCENES=abdeghil timeout -k 10 120 python3 tools/config_times.py >> $out 2>&1
  RT_NO_LEVELS_AHEAD=1 RT_PACKET_LEVELS=$L CONFIG=C3 timeout -k 10 120 python3 tools/config_times.py >> $out 2>&1
done
