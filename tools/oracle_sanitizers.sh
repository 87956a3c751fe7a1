# Runs the oracle's CPU tests against the ASan+UBSan and the TSan builds of oracle/smesh_oracle.cpp.
# The sanitizer runtimes must be preloaded because the host process is python.  usage: bash tools/oracle_sanitizers.sh
set -e
cd "$(dirname "$0")/.."
make -s -C oracle sanitize
asan=$(gcc -print-file-name=libasan.so); tsan=$(gcc -print-file-name=libtsan.so)
sel="tests/test_oracle.py tests/test_host.py"
echo "== ASan + UBSan"
SMESH_ORACLE_LIB=$PWD/oracle/_san/libsmesh_oracle_asan.so LD_PRELOAD=$asan ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  python -m pytest $sel -q -x -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
echo "== TSan (OpenMP add() with per-primitive locks, 4 threads)"
SMESH_ORACLE_LIB=$PWD/oracle/_san/libsmesh_oracle_tsan.so LD_PRELOAD=$tsan TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 ignore_noninstrumented_modules=1" OMP_NUM_THREADS=4 \
  python -m pytest tests/test_oracle.py -q -x -s -m "not gpu" -p no:cacheprovider -k "threaded or golden_cfg1_fuse or fast_histogram" > /tmp/smesh_tsan.log 2>&1 || true
tail -2 /tmp/smesh_tsan.log
# libgomp is not built with TSan annotations, so its barriers are invisible: every access of the main thread AFTER a parallel region
# is reported against the workers' accesses INSIDE it.  Those are false positives; a report between two worker threads would be real.
python3 - <<'PY'
import re
reps = open("/tmp/smesh_tsan.log").read().split("WARNING: ThreadSanitizer: data race")[1:]
real = [r for r in reps if not any("main thread" in l for l in [l for l in r.splitlines() if re.match(r"\s+(Write|Read|Previous write|Previous read|Atomic)", l)][:2])]
print("TSan reports: %d, of which between two OpenMP worker threads (real races): %d" % (len(reps), len(real)))
for r in real:
    print(r[:1200])
raise SystemExit(1 if real else 0)
PY
