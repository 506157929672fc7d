for i in 1 2; do
for v in "backbone,small,fuse,feat,stem,flow,nq" "backbone,small,fuse,feat,stem,flow,nq,conv3s"; do
echo "LSFA_OWN_CONV=$v"
LSFA_OWN_CONV=$v timeout 200 python tools/key_sections.py 2>/dev/null | grep -E "backbone|whole key|small net|whole non-key"
done; done
