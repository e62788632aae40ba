for cfg in "resnet basic" "segformer_attn_conv basic" "segformer_attn_conv projected_d,basic" "mobile_resnet_attn projected_d,basic"; do
  bash tools/ab_cut0_env.sh "$cfg" "JG_FORK_GAN=0" "JG_FORK_GAN=1" | head -2
done
