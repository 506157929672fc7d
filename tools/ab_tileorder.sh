# the (slice, channel tile, pixel tile) -> XCD mapping of the split convolution (LSFA_CONV_TILE_ORDER, see xcd_tile): key-frame sections
for i in 1 2; do for o in 0 1; do
echo "== LSFA_CONV_TILE_ORDER=$o"
LSFA_CONV_TILE_ORDER=$o timeout 200 python tools/key_sections.py 2>/dev/null | grep -E "backbone|flownet|whole key|small net|whole non-key"
done; done
