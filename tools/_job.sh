cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_host_operator.py tests/test_c1_plumbing.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -3
for z in 1 0 1 0; do
NL_UPLOAD_PULL=$z timeout 600 python bench.py --steps 3 --warmup 1 --preheat-steps 8 --no-cpu --no-also --apply > $O/apply_z$z.json 2> $O/apply_z$z.err
python - $O/apply_z$z.json $z <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        a=json.loads(l)["apply_from_host"]
        for k in ("fp32","fits_int16"):
            e=a[k]; print("pull",sys.argv[2],k,{x:e[x] for x in ("wall_ms","ms_upload_calls","ms_run","ms_upload_tail_in_run","ms_result_download","ms_destroy","upload_gib_s")}, "first:",e["first_apply_of_the_process"]["wall_ms"], e.get("result_equals_resident_pass"))
PY
done
G1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
G2="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH"
G3="SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
G4="SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
G5="SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
tools/pmc_probe.sh lf128 "$G1" "$G2" "$G3" "$G4" "$G5" -- --mode 5 > $O/lf128_pmc.txt 2>&1
tools/pmc_probe.sh lf32 "$G1" "$G2" "$G3" "$G4" "$G5" -- --mode 5 --frames 32 > $O/lf32_pmc.txt 2>&1
grep -A6 "linfit_fast_kernel<128, false>\|linfit_fast_kernel<32, false>" $O/lf128_pmc.txt $O/lf32_pmc.txt | cut -c1-200
