#!/bin/bash
# Sanitizer builds of the host side (SURVEY 5.2 counterpart).  usage: tools/sanitize/run.sh [gpu]
#   always: the helper pool of runtime.hip under ThreadSanitizer and Address+UB sanitizers, the mixed-radix clFFT planner under
#           Address+UB sanitizers (no GPU needed)
#   gpu:    the C++ block layer + CLI rebuilt with AddressSanitizer and run against the scheduler-contract, X-engine streaming and
#           per-block known-answer tests on the device (the HIP library itself stays uninstrumented)
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/tools/sanitize/build
mkdir -p $B
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
fail=0
for san in thread address,undefined; do
    out=$B/pool_test_${san%%,*}
    $HIPCC -O1 -g -std=c++17 --offload-arch=gfx950 -fsanitize=$san -fno-omit-frame-pointer -I$R/include -I$R/gr-clenabled_amd/csrc \
        $R/gr-clenabled_amd/csrc/runtime.hip $R/tools/sanitize/pool_test.cc -o $out -lpthread 2> $B/build_${san%%,*}.log || { echo "build failed ($san)"; tail -5 $B/build_${san%%,*}.log; fail=1; continue; }
    echo "== pool stress under -fsanitize=$san"
    TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ASAN_OPTIONS="detect_leaks=0" $out 2>&1 | tail -15 || fail=1
done
echo "== mixed-radix clFFT planner under -fsanitize=address,undefined"
$HIPCC -O1 -g -std=c++17 --offload-arch=gfx950 -fsanitize=address,undefined -fno-omit-frame-pointer -I$R/include -I$R/gr-clenabled_amd/csrc \
    $R/gr-clenabled_amd/csrc/runtime.hip $R/gr-clenabled_amd/csrc/fft_mr.hip $R/tools/sanitize/plan_test.cc -o $B/plan_test -lpthread 2> $B/build_plan.log \
    || { echo "build failed (planner)"; tail -5 $B/build_plan.log; fail=1; }
[ -x $B/plan_test ] && { ASAN_OPTIONS="detect_leaks=0" $B/plan_test 2>&1 | tail -5 || fail=1; }
if [ "${1:-}" = gpu ]; then
    echo "== block layer + CLI under AddressSanitizer (device run)"
    g++ -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -I$R/gr-clenabled_amd/host/include -I$R/include -shared \
        -o $B/libgnuradio-clenabled-mi355.so $R/gr-clenabled_amd/host/lib/clenabled_impl.cc -L$R/gr-clenabled_amd -lmi355_clenabled -Wl,-rpath,$R/gr-clenabled_amd || fail=1
    g++ -O1 -g -std=c++17 -fsanitize=address -fno-omit-frame-pointer -I$R/gr-clenabled_amd/host/include -I$R/include -o $B/test-clenabled-asan \
        $R/gr-clenabled_amd/host/apps/test_clenabled.cc -L$B -lgnuradio-clenabled-mi355 -L$R/gr-clenabled_amd -lmi355_clenabled \
        -Wl,-rpath,$B -Wl,-rpath,$R/gr-clenabled_amd -Wl,-rpath,/opt/rocm/lib || fail=1
    export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0"
    d=$(mktemp -d)
    for args in "--scheduler-contract" "--xengine-stream=$d" "--iterations=3"; do
        $B/test-clenabled-asan $args > $B/asan_run.log 2>&1
        rc=$?
        tail -3 $B/asan_run.log
        if [ $rc != 0 ] || grep -q "ERROR: AddressSanitizer" $B/asan_run.log; then fail=1; grep -A12 "ERROR: AddressSanitizer" $B/asan_run.log | head -20; fi
    done
    rm -rf $d
fi
[ $fail = 0 ] && echo "sanitizers: clean" || echo "sanitizers: FAILURES"
exit $fail
