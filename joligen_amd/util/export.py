"""Export of GAN generators (`BaseModel.export_networks`): mirror of /root/reference/util/export.py:1-45 and of the call site
/root/reference/models/base_model.py:870-938.

The reference exports by building its own CPU generator (`gan_networks.define_G(**vars(opt))`), loading the `<epoch>_net_G_A.pth` that
`save_networks` just wrote, and tracing it to ONNX (always) and TorchScript (`train_export_jit`).  The HIP modules of this package cannot
be traced -- they launch kernels through a C ABI -- so the exporter does what the reference does: it builds a plain-`torch.nn` module with
the SAME state_dict keys (the checkpoints interchange with the reference), loads the `.pth`, and traces THAT on the CPU.  These mirrors
are export artefacts only (an exported graph runs elsewhere, in ONNX Runtime / libtorch); no training or inference path of this package
uses them.

Mirrors exist for the generators the CUT path builds: `resnet_{n}blocks` / `resnet`
(models/modules/resnet_architecture/resnet_generator.py:167-347) and `resnet_attn` / `mobile_resnet_attn` (:350-557, mobile_modules.py:4-40,
attn_network.py:6-54).  `segformer_attn_conv`: the reference itself skips the ONNX export with torch 2 (base_model.py:905-909); its
TorchScript export needs the mmseg-style backbone and is reported as skipped.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _ResnetBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv_block = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), nn.InstanceNorm2d(dim), nn.ReLU(True),
                                        nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), nn.InstanceNorm2d(dim))

    def forward(self, x):
        return x + self.conv_block(x)


class _Seq(nn.Module):
    """`encoder.model.<i>` / `decoder.model.<i>`: the reference wraps each half in a module with a `.model` Sequential"""

    def __init__(self, *layers):
        super().__init__()
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class TorchResnetGenerator(nn.Module):
    """ResnetGenerator = ResnetEncoder + ResnetDecoder, InstanceNorm, reflect padding, no dropout (the configuration CUTModel builds)"""

    def __init__(self, input_nc, output_nc, ngf=64, n_blocks=9):
        super().__init__()
        enc = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, 7), nn.InstanceNorm2d(ngf), nn.ReLU(True),
               nn.Conv2d(ngf, ngf * 2, 3, stride=2, padding=1), nn.InstanceNorm2d(ngf * 2), nn.ReLU(True),
               nn.Conv2d(ngf * 2, ngf * 4, 3, stride=2, padding=1), nn.InstanceNorm2d(ngf * 4), nn.ReLU(True)]
        enc += [_ResnetBlock(ngf * 4) for _ in range(n_blocks)]
        dec = [nn.ConvTranspose2d(ngf * 4, ngf * 2, 3, stride=2, padding=1, output_padding=1), nn.InstanceNorm2d(ngf * 2), nn.ReLU(True),
               nn.ConvTranspose2d(ngf * 2, ngf, 3, stride=2, padding=1, output_padding=1), nn.InstanceNorm2d(ngf), nn.ReLU(True),
               nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7), nn.Tanh()]
        self.encoder, self.decoder = _Seq(*enc), _Seq(*dec)

    def forward(self, x):
        return self.decoder(self.encoder(x))


class _SeparableConv2d(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(c, c, 3, padding=1, padding_mode="reflect", groups=c), nn.InstanceNorm2d(c), nn.Conv2d(c, c, 1))

    def forward(self, x):
        return self.conv(x)


class _AttnBlock(nn.Module):
    def __init__(self, c, mobile):
        super().__init__()
        mk = (lambda: _SeparableConv2d(c)) if mobile else (lambda: nn.Conv2d(c, c, 3, padding=1, padding_mode="reflect"))
        self.conv1, self.conv1_norm = mk(), nn.InstanceNorm2d(c)
        self.conv2, self.conv2_norm = mk(), nn.InstanceNorm2d(c)

    def forward(self, x):
        h = F.relu(self.conv1_norm(self.conv1(x)))
        return x + self.conv2_norm(self.conv2(h))


class TorchResnetGeneratorAttn(nn.Module):
    """ResnetGenerator_attn + BaseGenerator_attn.forward: (nb_mask_attn - nb_mask_input) generated images and the input, blended by a
    softmax attention map"""

    def __init__(self, input_nc, output_nc, nb_mask_attn, nb_mask_input, ngf=64, n_blocks=9, mobile=False, twice_resnet_blocks=False):
        super().__init__()
        self.nb_mask_attn, self.nb_mask_input, self.output_nc, self.twice = nb_mask_attn, nb_mask_input, output_nc, twice_resnet_blocks
        self.conv1, self.conv1_norm = nn.Conv2d(input_nc, ngf, 7), nn.InstanceNorm2d(ngf)
        self.conv2, self.conv2_norm = nn.Conv2d(ngf, ngf * 2, 3, 2, 1), nn.InstanceNorm2d(ngf * 2)
        self.conv3, self.conv3_norm = nn.Conv2d(ngf * 2, ngf * 4, 3, 2, 1), nn.InstanceNorm2d(ngf * 4)
        self.resnet_blocks = nn.Sequential(*[_AttnBlock(ngf * 4, mobile) for _ in range(n_blocks)])
        n_img = nb_mask_attn - nb_mask_input
        self.deconv1_content, self.deconv1_norm_content = nn.ConvTranspose2d(ngf * 4, ngf * 2, 3, 2, 1, 1), nn.InstanceNorm2d(ngf * 2)
        self.deconv2_content, self.deconv2_norm_content = nn.ConvTranspose2d(ngf * 2, ngf, 3, 2, 1, 1), nn.InstanceNorm2d(ngf)
        self.deconv3_content = nn.Conv2d(ngf, output_nc * n_img, 7)
        self.deconv1_attention, self.deconv1_norm_attention = nn.ConvTranspose2d(ngf * 4, ngf * 2, 3, 2, 1, 1), nn.InstanceNorm2d(ngf * 2)
        self.deconv2_attention, self.deconv2_norm_attention = nn.ConvTranspose2d(ngf * 2, ngf, 3, 2, 1, 1), nn.InstanceNorm2d(ngf)
        self.deconv3_attention = nn.Conv2d(ngf, nb_mask_attn, 1)

    def forward(self, x):
        h = F.relu(self.conv1_norm(self.conv1(F.pad(x, (3, 3, 3, 3), mode="reflect"))))
        h = F.relu(self.conv2_norm(self.conv2(h)))
        h = F.relu(self.conv3_norm(self.conv3(h)))
        h = self.resnet_blocks(h)
        if self.twice:
            h = self.resnet_blocks(h)
        c = F.relu(self.deconv1_norm_content(self.deconv1_content(h)))
        c = F.relu(self.deconv2_norm_content(self.deconv2_content(c)))
        image = torch.tanh(self.deconv3_content(F.pad(c, (3, 3, 3, 3), mode="reflect")))
        a = F.relu(self.deconv1_norm_attention(self.deconv1_attention(h)))
        a = F.relu(self.deconv2_norm_attention(self.deconv2_attention(a)))
        att = torch.softmax(self.deconv3_attention(a), dim=1)
        nc, n_img = self.output_nc, self.nb_mask_attn - self.nb_mask_input
        out = image[:, 0:nc] * att[:, 0:1]
        for i in range(1, n_img):
            out = out + image[:, nc * i:nc * (i + 1)] * att[:, i:i + 1]
        for i in range(n_img, self.nb_mask_attn):
            out = out + x[:, :nc] * att[:, i:i + 1]
        return out


def define_torch_G(opt):
    """plain-torch generator with the reference's state_dict keys for opt.G_netG, or None when no mirror exists"""
    g = opt.G_netG
    if g in ("resnet", "resnet_9blocks", "resnet_6blocks"):
        return TorchResnetGenerator(opt.model_input_nc, opt.model_output_nc, opt.G_ngf, opt.G_nblocks)
    if g in ("resnet_attn", "mobile_resnet_attn"):
        return TorchResnetGeneratorAttn(opt.model_input_nc, opt.model_output_nc, getattr(opt, "G_attn_nb_mask_attn", 10),
                                        getattr(opt, "G_attn_nb_mask_input", 1), opt.G_ngf, opt.G_nblocks, mobile=g.startswith("mobile"),
                                        twice_resnet_blocks=getattr(opt, "G_backward_compatibility_twice_resnet_blocks", False))
    return None


def export(opt, model_in_file, model_out_file, opset_version=12, export_type="onnx"):
    """util/export.py:7-45: load the checkpoint into the CPU generator and trace it.  Returns the written path, or None with a printed
    reason when the export cannot be produced on this installation (no mirror for the generator; the `onnx` package missing)."""
    model = define_torch_G(opt)
    if model is None:
        print(f"[joligen_amd] export skipped: no plain-torch mirror of G_netG={opt.G_netG!r} (run the reference's util/export.py on {model_in_file})")
        return None
    model.eval()
    model.load_state_dict(torch.load(model_in_file, map_location="cpu"))
    dummy = torch.randn(1, opt.model_input_nc, opt.data_crop_size, opt.data_crop_size)
    if export_type == "onnx":
        try:
            torch.onnx.export(model, dummy, model_out_file, verbose=False, opset_version=opset_version, dynamo=False)
        except Exception as e:      # torch.onnx.errors.OnnxExporterError when the `onnx` package is not installed
            if "onnx" in str(e).lower() and "not installed" in str(e).lower():
                print(f"[joligen_amd] ONNX export skipped: {e}")
                return None
            raise
    elif export_type == "jit":
        with torch.no_grad():
            torch.jit.trace(model, dummy).save(model_out_file)
    else:
        raise ValueError(f"{export_type} is not available")
    return model_out_file
