# round 6: the persistent tail inside BATCHES (the round's first form ran it for wide root lumps whatever the batch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 16 4; do
echo "# batch of $b, ms per matrix: BSP_TAIL_IN_BATCH=1 (tail in the batch) | the product (second plan without it)"
python tools/ab_suite.py --reps=7 "--filter=^10|^20|^21|^31|^32|^12" --batch=$b "BSP_TAIL_IN_BATCH=1" - 2>&1 | grep -v "Warning\|amdgpu.ids" | sed 's/ lv [ 0-9]* ln [ 0-9]* r [0-9e.-]*//g'
done
