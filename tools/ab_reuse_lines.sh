for cfg in "resnet basic" "mobile_resnet_attn projected_d,basic"; do
  bash tools/ab_cut0_env.sh "$cfg" "JG_NCE_REUSE_FEATS=0" "JG_NCE_REUSE_FEATS=1" 2>/dev/null | head -3
done
