# VALU instructions and time of the match-statistics kernel with stages switched off (GENIE_OPT_SEARCH_STAGES_OFF)
for dbg in 0 1 3 7 15 47; do
  echo "== stages off mask $dbg"
  SEARCH_ONLY=1 DBG=$dbg bash tools/pmc_quick.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS TCP_TCC_READ_REQ_sum" 2>&1 | grep match_table
  RUNS="SEARCH_ONLY=1 DBG=$dbg" bash tools/ka_sweep.sh 2>&1 | tail -1
done
