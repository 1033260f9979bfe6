#!/bin/bash
# A/B of library builds on one of the other configurations: CONFIG=C4 tools/ab_config.sh lib1.so lib2.so ...
set -u
for lib in "$@"; do
  echo "== $lib"
  RT_LIB_PATH=$PWD/$lib timeout -k 10 300 python tools/config_times.py || exit 1
done
