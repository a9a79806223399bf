import ctypes, sys, subprocess, shlex
lib = ctypes.CDLL("/root/repo/geodesic_raytracing_amd/libgeodesic_hip.so")
name = sys.argv[1]
m = ctypes.c_void_p(); lib.gr_metric_builtin(name.encode(), ctypes.byref(m))
need = ctypes.c_size_t()
lib.gr_metric_argument_string(m, None, 0, None, 0, None, 0, ctypes.byref(need))
buf = ctypes.create_string_buffer(need.value)
lib.gr_metric_argument_string(m, None, 0, None, 0, buf, need.value, ctypes.byref(need))
args = buf.value.decode().split()
extra = sys.argv[2:]
cmd = ["/opt/rocm/lib/llvm/bin/clang++","--rocm-path=/opt/rocm","-include","hip/hip_runtime.h","--offload-arch=gfx950","-O3","-std=c++17","-ffp-contract=fast","-fno-math-errno","-freciprocal-math","-fassociative-math","-fno-signed-zeros","-fno-trapping-math","--cuda-device-only","-c","-x","hip","/root/repo/geodesic_raytracing_amd/csrc/kernels/geodesic_kernels.hip","-o","/tmp/k_%s.o"%name,"-Rpass-analysis=kernel-resource-usage"]+args+extra
r = subprocess.run(cmd, capture_output=True, text=True)
import re
cur=None
for line in r.stderr.splitlines():
    mm = re.search(r"Function Name: (\S+)", line)
    if mm: cur=mm.group(1)
    for key in ["VGPRs:", "SGPRs:", "ScratchSize", "Occupancy", "AGPRs"]:
        if key in line and "remark" in line:
            print(cur, line.split("remark:")[1].strip())
if r.returncode: print(r.stderr[-3000:])
