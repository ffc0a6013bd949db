# same-box breakdown of the scan kernel: full / no true hits / no candidates (MK_ABLATE=1) / loads+pack only (7)
run() { echo -n "$1: "; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])"; }
for r in 1 2; do
unset MERKURIO_LIB_PATH
run "full, 1% reads hit      "
run "full, no planted hits   " --plant-every 0
export MERKURIO_LIB_PATH=merkurio_amd/lib/libmerkurio_hip_abl8.so
run "ring only (ABLATE=8)    " --plant-every 0
export MERKURIO_LIB_PATH=merkurio_amd/lib/libmerkurio_hip_abl1.so
run "filter only (ABLATE=1)  "
export MERKURIO_LIB_PATH=merkurio_amd/lib/libmerkurio_hip_abl7.so
run "loads + pack (ABLATE=7) "
done
