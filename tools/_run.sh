cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/collect_round_profile.sh j 2>&1 | grep -E "value|busy|sra_fwd_wave|hbm_bytes_per_launch" | cut -c1-230
