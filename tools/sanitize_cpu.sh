#!/bin/bash
# CPU sanitizers (the pool has no GPU ASan): the host-logic harness (the headers the library shares with its kernels: layout, solvers,
# SE(3) maths) and the oracle, built with -fsanitize=address,undefined, under their own CPU tests.  Restores the normal builds afterwards.
cd "$(dirname "$0")/.." || exit 1
PRE=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
FLAGS="-O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-sanitize-recover=undefined -Wno-misleading-indentation"
python -c "import sys; sys.path.insert(0, 'oracle'); import pyoracle; pyoracle.build()" || exit 1
python -m pytest tests/test_host_logic.py -q -k layout > /dev/null 2>&1          # (builds tests/host_logic/libhost_logic.so)
cp tests/host_logic/libhost_logic.so /tmp/libhl_normal.so; cp oracle/libeds_oracle.so /tmp/liboracle_normal.so
g++ $FLAGS -o tests/host_logic/libhost_logic.so tests/host_logic/harness.cpp || exit 1
g++ $FLAGS -o oracle/libeds_oracle.so oracle/eds_oracle_capi.cpp || exit 1
LD_PRELOAD=$PRE ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_host_logic.py tests/test_oracle.py tests/test_oracle_properties.py -q -x
rc=$?
cp /tmp/libhl_normal.so tests/host_logic/libhost_logic.so; cp /tmp/liboracle_normal.so oracle/libeds_oracle.so
touch tests/host_logic/libhost_logic.so oracle/libeds_oracle.so
exit $rc
