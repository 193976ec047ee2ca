"""CPU tests of the int4 weight mode's oracle (oracle/int4_ref.py): the NF4 table against its published construction, the
comparison tree's tie rule, blockwise absmax, bf16 de-quantisation, and the two byte layouts (bitsandbytes' and the kernels')."""
import numpy as np
import torch

from oracle import int4_ref


def test_nf4_table_matches_its_published_construction():
    built = int4_ref.nf4_from_construction()
    tab = np.array(int4_ref.NF4_TABLE, dtype=np.float64)
    assert len(tab) == 16 and tab[0] == -1.0 and tab[7] == 0.0 and tab[15] == 1.0 and np.all(np.diff(tab) > 0)
    # torch.linspace / norm.ppf in fp32 inside bitsandbytes vs float64 here: the constants agree to a few fp32 ulps of the entries
    assert np.abs(built - tab).max() < 2e-6, np.abs(built - tab).max()
    assert np.array_equal(np.float32(tab).astype(np.float64), tab)                 # the constants ARE fp32 values
    # not symmetric: 8 positive levels (incl. 1.0) but 7 negative ones above -1.0 -> zero is exactly representable
    assert np.sum(tab > 0) == 8 and np.sum(tab < 0) == 7


def test_quantiser_tree_ties_blocks_and_bf16_dequantisation():
    mids = int4_ref.midpoints()
    assert abs(float(mids[14]) - 0.8614784181118011) < 1e-7 and abs(float(mids[7]) - 0.03979014977812767) < 1e-8   # dQuantizeNF4's constants
    t = int4_ref.table()
    w = torch.zeros(4, 128)
    w[0, :16] = t                                   # a block whose absmax is 1: every table entry maps to itself
    w[0, 64:80] = t * 0.5; w[0, 64] = -0.5          # absmax 0.5 in the row's second block
    w[1, 0] = 1.0; w[1, 1:16] = mids                # exactly on a midpoint -> the LOWER entry ( `x > m` is false )
    w[1, 64] = -2.0; w[1, 65] = 2.0 * 0.9           # 0.9 > 0.8615 -> code 15
    w[3, 5] = 3.0                                   # (row 2 stays all-zero: absmax 0)
    codes, absmax = int4_ref.quantize_blocks(w)
    assert absmax.shape == (4, 2) and absmax[0].tolist() == [1.0, 0.5] and absmax[2].tolist() == [0.0, 0.0]
    assert codes[0, :16].tolist() == list(range(16)) and codes[0, 64:80].tolist() == list(range(16))
    assert codes[1, 1:16].tolist() == list(range(15))
    assert codes[1, 64].item() == 0 and codes[1, 65].item() == 15
    dq = int4_ref.dequantize_blocks(codes, absmax)
    assert torch.equal(dq, dq.to(torch.bfloat16).float())                          # the int4 model is a bf16 model
    assert torch.equal(dq[2], torch.zeros(128)) and dq[3, 5] == 3.0
    assert torch.equal(dq[0, :16], t.to(torch.bfloat16).float())
    # Gaussian weights: the nearest-entry rule, checked by brute force; error bounded by half the widest gap times absmax
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(32, 256, generator=g) * 0.02).to(torch.bfloat16)
    codes, absmax = int4_ref.quantize_blocks(w)
    x = (w.float().reshape(32, 4, 64) * (1.0 / absmax).unsqueeze(-1)).reshape(32, 256)
    brute = (x.unsqueeze(-1) - t).abs().argmin(-1)
    assert (brute != codes.long()).float().mean() < 1e-3                           # (they differ only exactly on midpoints)
    err = (int4_ref.fake_quant(w) - w.float()).abs().reshape(32, 4, 64).amax(-1)
    assert bool((err <= absmax * (0.5 * 0.3039 + 2 ** -8)).all())                  # widest gap of the table: 1.0 - 0.6962


def test_byte_layouts():
    g = torch.Generator().manual_seed(1)
    codes = torch.randint(0, 16, (6, 192), generator=g).to(torch.uint8)
    bnb = int4_ref.pack_bnb(codes)
    assert bnb.numel() == codes.numel() // 2
    assert int(bnb[0]) == (int(codes[0, 0]) << 4) | int(codes[0, 1])               # even element in the high nibble
    pk = int4_ref.pack_kernel(codes)
    assert pk.shape == (6, 96)
    assert torch.equal(int4_ref.unpack_kernel(pk), codes)
    d0 = pk[0, :4].numpy().astype(np.uint32)
    x = int(d0[0] | (d0[1] << 8) | (d0[2] << 16) | (d0[3] << 24))                  # the first dword as the kernel loads it
    assert [(x >> (8 * j)) & 15 for j in range(4)] == codes[0, :4].tolist()        # x & 0x0f0f0f0f       = e0 e1 e2 e3
    assert [(x >> (8 * j + 4)) & 15 for j in range(4)] == codes[0, 4:8].tolist()   # (x >> 4) & 0x0f0f0f0f = e4 e5 e6 e7
