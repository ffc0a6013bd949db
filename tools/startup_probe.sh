# start-up floors of a run on the GPU box: the CLI doing nothing, a bare HIP program (runtime start, first allocation), loading the library
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do ( TIMEFORMAT="merkurio --version: %R s"; time merkurio_amd/lib/merkurio --version > /dev/null ); done 2>&1 | tail -3
cat > /tmp/hipinit.c <<'C'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
int main(){double t0=now();int n=0;hipGetDeviceCount(&n);double t1=now();void*p;hipMalloc(&p,1<<20);double t2=now();hipStream_t s;hipStreamCreate(&s);double t3=now();
printf("hipGetDeviceCount %.3f s, first hipMalloc %.3f s, stream %.3f s (devices %d)\n",t1-t0,t2-t1,t3-t2,n);return 0;}
C
/opt/rocm/bin/hipcc -x c++ -o /tmp/hipinit /tmp/hipinit.c 2>/dev/null
for i in 1 2 3; do ( TIMEFORMAT="  whole process: %R s"; time /tmp/hipinit ); done 2>&1
python3 - <<'P'
import ctypes, time, os
t=time.time(); L=ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"merkurio_amd/lib/libmerkurio_hip.so")); print("dlopen libmerkurio_hip.so (incl. libamdhip64): %.3f s"%(time.time()-t))
t=time.time(); n=L.mk_device_count(); print("mk_device_count: %.3f s"%(time.time()-t))
P
