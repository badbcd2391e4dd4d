set -e
mkdir -p gpurun_out
for c in c2 c4 mixed lorderf n96 n64 q200 q101 lorderf_lsoln; do
  for e in 1 0; do
    echo -n "epw=$e " >> gpurun_out/epw_cfg.log
    SAT_EXP_EPW=$e timeout -k 10 200 python scripts/run_config.py $c 3 2>&1 | tail -1 >> gpurun_out/epw_cfg.log
    SAT_EXP_EPW=$e python - <<'PY' >> gpurun_out/epw_cfg.log 2>&1 || true
PY
  done
done
cat gpurun_out/epw_cfg.log
