for extra in "" "SHL_MI355X_PATCH_PAIR=1" "SHL_MI355X_PATCH_PAIR=1 SHL_MI355X_PATCH=2,2,1" "SHL_MI355X_PATCH_PAIR=1 SHL_MI355X_PATCH=1,4,1" "SHL_MI355X_DEBUG=64" "SHL_MI355X_DEBUG_GEOM=1 SHL_MI355X_PATCH_PAIR=1"; do
echo "== $extra"
( export SHL_MI355X_IGEMM=patch SHL_EXPECT_KERNEL=patch SHL_EXPECT_FALLBACK=tile SHL_EXPECT_MIN=8 $extra
timeout 900 python -m pytest tests/forced_igemm_suite.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^patch geom" | tail -4 )
done
