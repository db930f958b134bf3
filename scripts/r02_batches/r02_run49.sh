cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q -rs 2>&1 | grep -i "skip" | head -5 &
wait
GRAPHS="googleplus pokec ogbl_ppa" bash scripts/ab_variants.sh cur plain 2>&1 | cut -c1-200
GRAPHS="orkut" FLAGS="4" bash scripts/ab_variants.sh cur plain 2>&1 | cut -c1-200
