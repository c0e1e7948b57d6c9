#!/bin/bash
# (fault_hunt5.sh restricted to the groups whose code changed after the whole-suite run: losses, legacy networks, the step.)
# The whole GPU suite with every tensor ENDING at an unmapped page (tools/dbg/guard_alloc.cpp, GUARD_ALLOC_END=1 GUARD_ALLOC_LEAK=1),
# kernels serialized, every library launch named (tuning build): an access >= 16 bytes past the end of ANY buffer aborts the run at the
# offending launch.  One pytest process per test file group; a group that aborts is re-run from the test after the culprit.
OUT=$PWD/gpurun_out/fault_hunt
mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 -w -shared -fPIC -o /tmp/libguard_alloc.so tools/dbg/guard_alloc.cpp || exit 1
export FDGAN_TEST_GUARD_ALLOC=/tmp/libguard_alloc.so GUARD_ALLOC_END=1 GUARD_ALLOC_LEAK=1 AMD_SERIALIZE_KERNEL=3 FDGAN_TEST_HYGIENE=none
export FDGAN_LIB=$PWD/fd-gan_amd/fdgan_hip/variants/libfdgan_hip_tune.so FDGAN_DEBUG_TRACE_LAUNCH=1
: > $OUT/summary6.txt
run() {
  local name=$1; shift
  timeout ${GUARD_TIMEOUT:-1200} python -X faulthandler -m pytest "$@" -m gpu -q -s -p no:cacheprovider > $OUT/$name.out 2> $OUT/$name.err
  local rc=$?
  echo "$name rc=$rc $(tail -1 $OUT/$name.out | cut -c1-150)" | tee -a $OUT/summary6.txt
  if [ $rc -ge 124 ]; then
    grep -a -n "\[launch\]\|Memory access fault" $OUT/$name.err | tail -4 | tee -a $OUT/summary6.txt
    grep -a -n "File \"" $OUT/$name.err | grep -v "site-packages\|dist-packages\|/usr/lib" | head -10 | tee -a $OUT/summary6.txt
  else
    grep -n "^FAILED" $OUT/$name.out | cut -c1-160 | tee -a $OUT/summary6.txt
  fi
  tail -c 100000 $OUT/$name.err > $OUT/$name.err.tail; rm -f $OUT/$name.err
}
SKIP="not rccl and not plan_replay and not graph"
run H3_losses tests/test_hip_losses.py -k "$SKIP"
run H5_models_legacy tests/test_hip_models.py -k "legacy or dehaze22 or pyramid"
run H6_models_step tests/test_hip_models.py -k "(full_size or training_step) and not trajectory"
