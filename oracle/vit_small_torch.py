"""`vit_small_patch16_224` as a plain torch module with timm's attribute names (TEST INFRASTRUCTURE ONLY).

The reference's projected discriminator with `D_proj_network_type = "vitsmall"` (what examples/example_gan_mario2sonic.json selects) builds
its frozen feature network with `timm.create_model("vit_small_patch16_224", img_size=D_proj_interp, pretrained=True)`
(/root/reference/models/modules/projected_d/projector.py:252-253,327-331) and reads the token sequences behind blocks 2, 5, 8 and 11 through
`configure_get_feats_vit_timm` (:138-153), which only touches `patch_embed`, `cls_token`, `pos_embed`, `pos_drop` and `blocks`.
timm (requirements.txt: timm==1.0.9) is an ABSENT dependency and its checkpoints cannot be downloaded here, so this file restates timm's
published definition of that model name -- vision_transformer.py `vit_small_patch16_224` (patch 16, width 384, depth 12, 6 heads, MLP
ratio 4, qkv bias, LayerNorm eps 1e-6, exact GELU, class token, learned position embedding, no pre-norm, no layer scale, final `norm` and
a 1000-way `head` that the projector never calls but whose entries are part of the state_dict), `Block` (x + attn(norm1(x)); x +
mlp(norm2(x))), `Attention` (packed qkv Linear, [3, heads, head_dim] channel order, scale head_dim ** -0.5), `Mlp` (fc1, GELU, fc2) and
layers/patch_embed.py `PatchEmbed` (Conv2d k = s = 16, flatten to [B, N, C]) -- with timm's attribute names, so that
oracle/make_golden_projd_vit.py can drive the UNMODIFIED reference `ProjectedDiscriminator("vitsmall")` through a stubbed
`timm.create_model`.  Parity of the backbone against timm itself is UNPINNED (no timm to run); what the fixture pins is the product's
HIP implementation of this architecture and everything downstream of it (Conv1d CCM, FeatureFusionBlockVector CSM, the MLP heads).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

WIDTH, DEPTH, HEADS, PATCH, MLP_RATIO, LN_EPS, NUM_CLASSES = 384, 12, 6, 16, 4, 1e-6, 1000


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch=PATCH, width=WIDTH):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.grid_size = (img_size // patch, img_size // patch)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(3, width, patch, stride=patch)
        self.norm = nn.Identity()

    def forward(self, x):
        assert x.shape[-2:] == self.img_size, (x.shape, self.img_size)      # timm asserts the input size as well
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class Attention(nn.Module):
    def __init__(self, dim=WIDTH, heads=HEADS):
        super().__init__()
        self.num_heads, self.head_dim = heads, dim // heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.q_norm, self.k_norm = nn.Identity(), nn.Identity()
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class Mlp(nn.Module):
    def __init__(self, dim=WIDTH, hidden=WIDTH * MLP_RATIO):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.drop1 = nn.Dropout(0.0)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop2 = nn.Dropout(0.0)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class Block(nn.Module):
    def __init__(self, dim=WIDTH, heads=HEADS):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=LN_EPS)
        self.attn = Attention(dim, heads)
        self.ls1, self.drop_path1 = nn.Identity(), nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=LN_EPS)
        self.mlp = Mlp(dim, dim * MLP_RATIO)
        self.ls2, self.drop_path2 = nn.Identity(), nn.Identity()

    def forward(self, x):
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        return x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))


class VitSmallPatch16(nn.Module):
    """timm.models.vision_transformer.VisionTransformer as configured by `vit_small_patch16_224`, created with `img_size`"""

    def __init__(self, img_size=224, width=WIDTH, depth=DEPTH, heads=HEADS):
        super().__init__()
        self.num_classes, self.embed_dim = NUM_CLASSES, width
        self.patch_embed = PatchEmbed(img_size, PATCH, width)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, width))
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + 1, width) * 0.02)
        self.pos_drop = nn.Dropout(0.0)
        self.patch_drop, self.norm_pre = nn.Identity(), nn.Identity()
        self.blocks = nn.Sequential(*[Block(width, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(width, eps=LN_EPS)
        self.fc_norm, self.head_drop = nn.Identity(), nn.Dropout(0.0)
        self.head = nn.Linear(width, NUM_CLASSES)
        nn.init.normal_(self.cls_token, std=1e-6)

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        x = self.pos_drop(x + self.pos_embed)
        return self.norm(self.blocks(self.norm_pre(self.patch_drop(x))))

    def forward(self, x):
        return self.head(self.head_drop(self.fc_norm(self.forward_features(x)[:, 0])))


def exact_gelu(x):
    return F.gelu(x)
