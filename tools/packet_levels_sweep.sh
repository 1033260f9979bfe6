# deeper levels of the ray tree through the packet kernels (RT_PACKET_LEVELS > 1): only the host-synchronised level loop can do
# that (the packet hand-out needs the exact ray count), so both settings run with RT_NO_LEVELS_AHEAD=1
set -e
mkdir -p gpurun_out/pl
out=gpurun_out/pl/times.txt
: > $out
for L in 1 2 8; do
  echo "== RT_NO_LEVELS_AHEAD=1 RT_PACKET_LEVELS=$L" >> $out
  RT_NO_LEVELS_AHEAD=1 RT_PACKET_LEVELS=$L CONFIG=C1 C1_SCENES=abdeghil timeout -k 10 120 python3 tools/config_times.py >> $out 2>&1
  RT_NO_LEVELS_AHEAD=1 RT_PACKET_LEVELS=$L CONFIG=C3 timeout -k 10 120 python3 tools/config_times.py >> $out 2>&1
done
