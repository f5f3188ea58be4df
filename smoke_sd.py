"""smoke() leg for the SD side (imported by __graft_entry__.smoke only — it uses the oracle as the checker, so it lives
outside the product package): one tiny denoised frame (reduced-width VAE/UNet of the SD architecture, 2 DDIM steps)
through the C ABI on cuda:0, checked against the CPU oracle."""
import torch


def run():
    from sd_video_gen_amd import _lib
    from oracle import sd_oracle as SO
    ctx = _lib.default_context()
    ucfg = dict(block_out=(64, 128), layers=1, heads=4, ctx_dim=64, groups=32, in_ch=4, out_ch=4, attn=(1, 0))
    vcfg = dict(block_out=(64, 128, 128, 128), layers=1, groups=32, latent=4)
    usd = SO.seeded_weights(SO.unet_shapes(ucfg), 5)
    vsd = SO.seeded_weights(SO.vae_shapes(vcfg), 6)
    ctx.configure(_lib.SVG_UNET, block_out=list(ucfg["block_out"]), layers=1, heads=4, ctx_dim=64, groups=32, attn=list(ucfg["attn"]), f16=1)
    ctx.load_state_dict(_lib.SVG_UNET, usd)
    ctx.finalize(_lib.SVG_UNET)
    ctx.configure(_lib.SVG_VAE, block_out=list(vcfg["block_out"]), layers=1, groups=32, latent=4, f16=1)     # fp16 storage: the default mode of the façade
    ctx.load_state_dict(_lib.SVG_VAE, vsd)
    ctx.finalize(_lib.SVG_VAE)
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (1, 64, 64, 3), dtype=torch.uint8, generator=g)
    eps = torch.randn(1, 4, 8, 8, generator=g)
    emb = torch.randn(2, 7, 64, generator=g)
    noise = torch.randn(1, 4, 8, 8, generator=g)

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())
    z = ctx.vae_encode(img.cuda(), eps=eps.cuda())
    z_ref = SO.encode_img(vsd, img, eps, vcfg)
    den = ctx.ddim_loop(z, emb.cuda(), num_steps=50, start_step=48, guidance=0.0, noise=noise.cuda())
    den_ref = SO.gen_i2i_latents(usd, emb, z_ref, 50, 0.0, 48, noise=noise, cfg=ucfg)
    frame = ctx.vae_decode(den).cpu()
    frame_ref = SO.decode_img_latents(vsd, den_ref, vcfg)
    e1, e2 = rel(z.cpu(), z_ref), rel(den.cpu(), den_ref)
    d = (frame.int() - frame_ref.int()).abs().float()
    assert e1 < 1e-2 and e2 < 1e-2 and d.mean() <= 0.5, (e1, e2, float(d.mean()))
    print("smoke: VAE enc rel-L2 %.2e, DDIM(2 steps) rel-L2 %.2e, frame mean |diff| %.3f LSB vs oracle" % (e1, e2, float(d.mean())))
