# pwdw_fused: dynamic instructions per wave up to each ablation point (SHL_MI355X_DEBUG 256: loads + MFMAs, 512: + partial sums in LDS,
# 1024: + pointwise epilogue, 0: whole kernel)
export TMPDIR=/tmp; R=$PWD; cd /tmp
for dbg in 256 512 1024 0; do
  export SHL_MI355X_DEBUG=$dbg
  rm -rf /tmp/pmc_ph
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_ph -o t -- python $R/bench.py --steps-only --steps 4 --warmup 1 --windows 1 > /dev/null 2>&1
  echo "== debug $dbg"; (cd $R; python tools/pmc_kernel_counters.py /tmp/pmc_ph | grep -E "^kernel|pwdw_fused|stemdw" | cut -c1-250)
done
