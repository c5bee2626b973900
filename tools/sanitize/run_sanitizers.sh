#!/bin/bash
# tools/sanitize/run_sanitizers.sh — every sanitizer run of the HOST code, on the CPU build, in this container (no GPU: GPU ASan and XNACK are
# not available on this pool and are not attempted).  Writes profiles/r06_sanitizers.txt.  ~12 minutes, most of it the instrumented build.
#   1. oracle/oracle.c under ASan + UBSan (make -C oracle asan), the whole `-m "not gpu"` suite against it
#   2. csrc/gate_registry.hpp (the init gate's slot registry) under TSan and under ASan + UBSan: 8 host threads, stub HIP entry points
#   3. libdmxq.so's host code under ASan + UBSan (build_host_asan.py): tests/test_abi_and_host.py and host_driver.py (27 k calls over every
#      plan boundary and descriptor-array size) through ctypes
#   4. csrc/torch_binding.cpp under ASan + UBSan: meta kernels and the CPU-tensor refusals of every op class (torch_binding_driver.py)
set -u -o pipefail
cd "$(dirname "$0")/../.."
OUT=profiles/r06_sanitizers.txt
ASANRT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
GCCASAN=$(gcc -print-file-name=libasan.so)
SO=$PWD/tools/sanitize/_out
fail=0
{
echo "# tools/sanitize/run_sanitizers.sh   commit $(git rev-parse HEAD)$(git diff --quiet || echo ' + uncommitted changes')   $(date -u +%Y-%m-%dT%H:%MZ)"
echo "# gcc $(gcc -dumpversion), clang $(/opt/rocm/lib/llvm/bin/clang++ --version | head -1 | sed 's/.*version //')"
echo
echo "== 1. oracle/oracle.c: -fsanitize=address,undefined (gcc), pytest -m 'not gpu' against oracle/_asan/liboracle_asan.so"
make -s -C oracle asan || fail=1
LD_PRELOAD=$GCCASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  ORACLE_LIB_PATH=$PWD/oracle/_asan/liboracle_asan.so python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3 || fail=1
LD_PRELOAD=$GCCASAN ASAN_OPTIONS=detect_leaks=0 ORACLE_LIB_PATH=$PWD/oracle/_asan/liboracle_asan.so python -c "
import sys; sys.path.insert(0, 'oracle'); import oracle; oracle.lib()
print('loaded:', [l.split()[-1] for l in open('/proc/self/maps') if 'liboracle' in l][0])" || fail=1
echo
echo "== 2. csrc/gate_registry.hpp: tools/sanitize/gate_registry_harness.cpp"
for mode in thread address,undefined; do
  g++ -std=c++17 -g -O1 -fsanitize=$mode -fno-sanitize-recover=undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/sanitize/gate_registry_harness.cpp \
      -o /tmp/gate_$$ -ldl -lpthread 2>&1 | grep -v "^$" | head -5
  echo "-fsanitize=$mode:"
  ASAN_OPTIONS=detect_leaks=0 TSAN_OPTIONS=halt_on_error=1 /tmp/gate_$$ || fail=1
done
rm -f /tmp/gate_$$
echo
echo "== 3. libdmxq.so host code: hipcc -fsanitize=address,undefined -fno-gpu-sanitize (tools/sanitize/build_host_asan.py)"
python tools/sanitize/build_host_asan.py > /dev/null || fail=1
E="LD_PRELOAD=$ASANRT ASAN_OPTIONS=detect_odr_violation=0:detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 DMXQ_LIB_PATH=$SO/libdmxq_asan.so"
env $E DMXQ_BINDING=ctypes python -m pytest tests/test_abi_and_host.py -q -p no:cacheprovider 2>&1 | tail -2 || fail=1
env $E DMXQ_BINDING=ctypes python tools/sanitize/host_driver.py 2>&1 || fail=1
echo
echo "== 4. csrc/torch_binding.cpp: clang++ -fsanitize=address,undefined, linked against the instrumented libdmxq"
python - <<'PY' || fail=1
import importlib.util, subprocess
spec = importlib.util.spec_from_file_location("_b", "dmx-compressor_amd/build.py"); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
inc, tlib = B._torch_flags()
out = "tools/sanitize/_out"
cc = "/opt/rocm/lib/llvm/bin/clang++"
# (-asan-globals=0: the string literals of libstdc++'s headers exist in both instrumented objects and would be reported as ODR violations)
subprocess.check_call([cc, "-O1", "-g", "-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-D_GLIBCXX_USE_CXX11_ABI=1", "-Wno-deprecated-declarations",
                       "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-sanitize=vptr", "-shared-libsan", "-fno-omit-frame-pointer",
                       "-mllvm", "-asan-globals=0"] + inc + ["-c", B.CSRC + "/torch_binding.cpp", "-o", out + "/torch_binding_asan.o"])
subprocess.check_call([cc, "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", out + "/dmxq_torch_asan.so", out + "/torch_binding_asan.o",
                       "-L" + out, "-ldmxq_asan", "-Wl,-rpath," + out, "-L" + tlib, "-Wl,-rpath," + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-ltorch_python"])
PY
env $E DMXQ_TORCH_LIB_PATH=$SO/dmxq_torch_asan.so python tools/sanitize/torch_binding_driver.py 2>&1 | tail -3 || fail=1
echo
if [ $fail = 0 ]; then echo "RESULT: clean (no sanitizer report, every step exited 0)"; else echo "RESULT: FAILURES above"; fi
} 2>&1 | tee $OUT
# (the instrumented objects are ~770 MB: a gpurun snapshot must stay under 512 MiB)
rm -rf tools/sanitize/_out oracle/_asan
