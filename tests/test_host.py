"""CPU tests of the host logic: flag surface, scope/variable semantics, parameter manifest parity
with the oracle, C-ABI symbol coverage (no compute calls - there is no GPU here)."""
import json
import os
import re

import numpy as np
import pytest
import torch

import biggan_tensorflow_amd  # noqa: F401
from biggan_tensorflow_amd import hip, main as M, model, scope as S, utils
from biggan_tensorflow_amd import DiffAugment as DA
from oracle import ref_model as RM
from oracle import ref_ops as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------- flag surface (main.py:9-147)
def test_flag_surface_matches_golden_listing():
    """tests/golden/flags.json lists the reference's 119 flags (name, type, default); it is data
    extracted once by tests/golden/make_flags.py, not reference source."""
    ref = json.load(open(os.path.join(GOLDEN, "flags.json")))
    assert len(ref) == 119
    mine = {n: (t.__name__, d) for n, t, d in M.FLAGS}
    assert len(mine) == 119
    for name, typ, default in ref:
        assert name in mine, name
        assert mine[name][0] == typ, (name, typ, mine[name][0])
        assert mine[name][1] == default, (name, default, mine[name][1])


def test_defaults_and_str2bool_leniency():
    a = M.parse_args([], make_dirs=False)
    assert a.gan_type == "ra-dragan" and a.img_size == 256 and a.ch == 64 and a.batch_size == 16
    assert a.sn is True and a.g_regularization == "ortho_cosine" and a.da_policy == "full"
    # utils.py:173-174 is a substring test
    assert utils.str2bool("") and utils.str2bool("t") and utils.str2bool("rue") and utils.str2bool("TRUE")
    assert not utils.str2bool("false") and not utils.str2bool("yes")
    assert M.parse_args(["--sn", "false"], make_dirs=False).sn is False


def test_check_args_creates_folders(tmp_path):
    d = [str(tmp_path / n) for n in ("c", "r", "l", "s")]
    M.parse_args(["--checkpoint_dir", d[0], "--result_dir", d[1], "--log_dir", d[2], "--sample_dir", d[3]])
    assert all(os.path.isdir(x) for x in d)


def test_class_conditional_manifest_on_cpu():
    """--n_labels widens first/dense1 and every cond-BN FC by n_labels and adds DC_logit without SN
    (BigGAN.py:346-365, 433, 689-701); built on the meta device, no GPU needed."""
    g = model.BigGAN(M.parse_args(["--gan_type", "hinge", "--img_size", "128", "--ch", "8", "--n_labels", "10",
                                   "--virtual_batches", "2"], make_dirs=False),
                     device="cpu", store=S.VariableStore("cpu"))
    g.build_model()
    v = g.store.vars
    assert tuple(v["generator/first/dense1/kernel"].shape) == (106, 200)          # (96+10), round_up(106*1.85, 8)
    assert tuple(v["generator/resblock_up_16/res1/batch_norm/beta/kernel"].shape)[0] == 42
    assert tuple(v["discriminator/DC_logit/kernel"].shape) == (128, 10)
    assert "discriminator/DC_logit/u" not in v and "discriminator/D_logit/u" in v
    assert g.virtual_batches == 2


def test_checkpoint_roundtrip_on_cpu(tmp_path):
    """save / load (BigGAN.py:1255-1284): variables, EMA shadows, Adam slots and step counters, keyed by
    the TF variable names, survive a round trip through safetensors (no GPU needed)."""
    def make():
        g = model.BigGAN(M.parse_args(["--gan_type", "hinge", "--img_size", "64", "--ch", "8"], make_dirs=False),
                         device="cpu", store=S.VariableStore("cpu", seed=3))
        return g.build_model()
    a = make()
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for arena in a.store.arenas.values():
            for buf in (arena.params, arena.m, arena.v) + ((arena.ema,) if arena.ema is not None else ()):
                buf.copy_(torch.randn(buf.shape, generator=gen))
        a.store.vars["generator/first/dense1/u"].copy_(torch.randn(1, a.store.vars["generator/first/dense1/u"].shape[1],
                                                                   generator=gen))
    a.counter, a.d_arena.step, a.g_arena.step = 17, 17, 16
    path = a.save(str(tmp_path), 17)
    assert path.endswith("BigGAN.model-17.safetensors") and os.path.exists(os.path.join(os.path.dirname(path), "checkpoint"))
    from safetensors import safe_open
    with safe_open(path, "pt") as f:
        keys = set(f.keys())
    assert "generator/first/dense1/kernel/ExponentialMovingAverage" in keys
    assert "discriminator/D_logit/kernel/Adam_1" in keys and "generator/first/dense1/u" in keys
    b = make()
    ok, counter = b.load(str(tmp_path))
    assert ok and counter == 17 and b.counter == 17 and (b.d_arena.step, b.g_arena.step) == (17, 16)
    sa, sb = a.state_tensors(), b.state_tensors()
    assert sa.keys() == sb.keys()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    c = make()
    assert c.load(str(tmp_path / "nowhere")) == (False, 0)


def test_abi_rejects_null_tensors_before_any_launch():
    """Every int-returning entry point, given valid sizes / descriptors but NULL tensor pointers, must return
    BG_ERR_ARG (1) from its argument validation - i.e. before it tries to launch anything (a launch attempt
    would return BG_ERR_LAUNCH = 2 here, where there is no GPU; on a GPU it would fault)."""
    import ctypes
    L = hip.lib()
    cd = hip.conv_desc(2, 8, 8, 16, 8, 8, 32, 3, 1, 1, 0)
    dd = hip.conv_desc(2, 8, 8, 16, 16, 16, 32, 4, 2, 1, 1)
    rd = hip.conv_desc(2, 8, 8, 16, 8, 8, 3, 3, 1, 1, 0)
    td = hip.conv_desc(2, 8, 8, 3, 4, 4, 32, 3, 2, 1, 0)
    gd = hip.BgGemmDesc(128, 128, 128, 0, 0, 128, 128, 128, 1, 0, 0, 0)
    checked = 0
    for name, (res, args) in hip.SIGNATURES.items():
        if res is not ctypes.c_int or "supported" in name or name in ("bg_abi_version",
                                                                      "bg_prof_collect", "bg_prof_dump"):
            continue
        vals = []
        for a in args:
            if a is hip._CD:
                vals.append(dd if "deconv" in name else rd if "rgb" in name else td if "thin" in name else cd)
            elif a is hip._GD:
                vals.append(gd)
            elif a is hip._AD:
                ad = hip.BgAttn16Desc()
                ad.B, ad.N, ad.Nk, ad.d, ad.dv = 1, 128, 128, 16, 32
                for f_ in ("ldq", "ldk", "lddq", "lddk"):
                    setattr(ad, f_, 16)
                for f_ in ("ldv", "ldo", "ldg", "lddv"):
                    setattr(ad, f_, 32)
                vals.append(ad)
            elif a is ctypes.c_void_p or a is ctypes.c_char_p:
                vals.append(None)
            elif a is ctypes.POINTER(hip.BgDenseItem):
                items = (hip.BgDenseItem * 2)()          # valid sizes, NULL tensors
                for it in items:
                    it.K, it.N, it.ldx = 32, 64, 32
                vals.append(items)
            elif a in (ctypes.c_float, ctypes.c_double):
                vals.append(1.0)
            elif a is ctypes.c_size_t:
                vals.append(1 << 20)
            else:
                vals.append(2 if "dense_group" in name and a is ctypes.c_int and len(vals) == 1 else 128)
        rc = getattr(L, name)(*vals)
        assert rc == 1, (name, rc, L.bg_last_error())
        checked += 1
    assert checked >= 50


def test_sample_grid_png(tmp_path):
    """utils.save_images (utils.py:133-161): [-1,1] images tiled row-major into a grid, 8-bit PNG."""
    import struct
    import zlib
    imgs = np.zeros((4, 2, 3, 3), np.float32) - 1.0
    for i in range(4):
        imgs[i, :, :, i % 3] = 1.0 if i < 3 else 0.0
    path = utils.save_images(imgs, [2, 2], str(tmp_path / "grid.png"))
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    w, h, depth, ctype = struct.unpack(">IIBB", data[16:26])
    assert (w, h, depth, ctype) == (6, 4, 8, 2)
    n = struct.unpack(">I", data[33:37])[0]
    assert data[37:41] == b"IDAT"
    raw = zlib.decompress(data[41:41 + n])
    rows = np.frombuffer(raw, np.uint8).reshape(4, 1 + 6 * 3)[:, 1:].reshape(4, 6, 3)
    assert tuple(rows[0, 0]) == (255, 0, 0) and tuple(rows[0, 3]) == (0, 255, 0)      # images 0, 1 on the first row
    assert tuple(rows[2, 0]) == (0, 0, 255) and tuple(rows[2, 3]) == (128, 0, 0)      # images 2, 3 below (0.0 -> 128)
    assert utils.merge(imgs, [2, 2]).shape == (4, 6, 3)


def test_out_of_scope_flags_rejected_at_build():
    for extra in (["--g_final_layer", "true"], ["--cls_embedding", "true"], ["--z_reconstruct", "true"]):
        argv = ["--gan_type", "hinge", "--img_size", "64"] + extra
        with pytest.raises(NotImplementedError):
            model.BigGAN(M.parse_args(argv, make_dirs=False), device="cpu", store=S.VariableStore("cpu"))
    # (round 2: the gradient penalty with --bn_in_d is built - tests/test_gpu_step.py - and no longer rejected)
    model.BigGAN(M.parse_args(["--gan_type", "ra-dragan", "--img_size", "64", "--bn_in_d", "true"], make_dirs=False),
                 device="cpu", store=S.VariableStore("cpu"))
    with pytest.raises(ValueError):
        g = model.BigGAN(M.parse_args(["--gan_type", "hinge", "--img_size", "96"], make_dirs=False), device="cpu",
                         store=S.VariableStore("cpu"))
        g.generator(torch.empty(2, 1, 1, 256, device="meta"))


# ---------------------------------------------------------------- scope semantics
def test_variable_scope_names_and_default_name_uniquifying():
    st = S.VariableStore("cpu")
    with st.variable_scope("discriminator"):
        with st.variable_scope("res1"):
            with st.variable_scope(None, default_name="prelu") as a:
                pass
            with st.variable_scope(None, default_name="prelu") as b:
                pass
        with st.variable_scope("res1"):                       # re-entered: counters were cleared
            with st.variable_scope(None, default_name="prelu") as c:
                pass
    assert a == "discriminator/res1/prelu" and b == "discriminator/res1/prelu_1" and c == a
    with st.variable_scope("g"):
        v = st.get_variable("kernel", [3, 4])
        assert st.get_variable("kernel", [3, 4]) is v
        with pytest.raises(ValueError):
            st.get_variable("kernel", [4, 4])
    assert v.bg_name == "g/kernel" and v.requires_grad


def test_initialisers():
    rng = np.random.default_rng(0)
    a = S.truncated_normal_initializer(0.0, 0.02)((1000, 50), rng)
    assert np.abs(a).max() <= 0.04 + 1e-9 and abs(a.std() - 0.02 * 0.88) < 2e-3
    assert S.constant_initializer(1.0)((3,), rng).tolist() == [1, 1, 1]


# ---------------------------------------------------------------- manifest == oracle manifest
@pytest.mark.parametrize("size", [64, 128, 256, 512])
def test_manifest_matches_oracle(size):
    args = M.parse_args(["--gan_type", "hinge", "--img_size", str(size), "--ch", "8"], make_dirs=False)
    store = S.VariableStore("cpu")
    gan = model.BigGAN(args, device="cpu", store=store)
    img = gan.generator(torch.empty(2, 1, 1, gan.z_dim, device="meta"))
    out = gan.discriminator(img)
    assert tuple(img.shape) == (2, size, size, 3) and tuple(out["real"].shape) == (2, 1)
    tr = RM.Trainer(RM.Config(img_size=size, ch=8, batch_size=2), torch.float32).build()
    mine = {k: tuple(v.shape) for k, v in store.vars.items()}
    ref = {k: tuple(v.shape) for k, v in tr.vs.vars.items()}
    assert mine == ref
    assert {k for k in mine if store.trainable[k]} == {k for k in ref if tr.vs.trainable[k]}
    # second instantiation resolves the same names (eager re-execution == reuse)
    n = len(store.vars)
    gan.generator(torch.empty(2, 1, 1, gan.z_dim, device="meta"))
    gan.discriminator(img)
    assert len(store.vars) == n


@pytest.mark.parametrize("shared", [False, True])
def test_batch_renorm_manifest_matches_oracle(shared):
    """--bn_type batch_renorm (ops.py:556-559, 573-576, 600-609, 645-715): the 'batch_renorm' scopes, the extra running
    statistics of the non-shared form and the Keras layer's renorm_mean / renorm_stddev under --bn_in_d."""
    argv = ["--gan_type", "hinge", "--img_size", "64", "--ch", "8", "--bn_type", "batch_renorm", "--bn_in_d", "true",
            "--bn_renorm_shared", str(shared).lower()]
    store = S.VariableStore("cpu")
    gan = model.BigGAN(M.parse_args(argv, make_dirs=False), device="cpu", store=store)
    gan.discriminator(gan.generator(torch.empty(2, 1, 1, gan.z_dim, device="meta")))
    tr = RM.Trainer(RM.Config(img_size=64, ch=8, batch_size=2, bn_type="batch_renorm", bn_in_d=True,
                              bn_renorm_shared=shared), torch.float32).build()
    mine = {k: tuple(v.shape) for k, v in store.vars.items()}
    ref = {k: tuple(v.shape) for k, v in tr.vs.vars.items()}
    assert mine == ref
    assert {k for k in mine if store.trainable[k]} == {k for k in ref if tr.vs.trainable[k]}
    assert not any(k.endswith("/batch_norm/pop_mean") for k in mine)
    assert ("generator/resblock_up_8/res1/batch_renorm/renorm_weight" in mine) == (not shared)
    assert "discriminator/resblock_down_1/res1/batch_renorm/renorm_stddev" in mine


def test_arena_packing_on_cpu():
    args = M.parse_args(["--gan_type", "hinge", "--img_size", "64", "--ch", "8", "--z_dim", "64"], make_dirs=False)
    store = S.VariableStore("cpu")
    gan = model.BigGAN(args, device="cpu", store=store)
    gan.discriminator(gan.generator(torch.empty(2, 1, 1, 64, device="meta")))
    before = store.export_arrays()
    store.pack()
    after = store.export_arrays()
    assert all(np.array_equal(before[k], after[k]) for k in before)
    ga = store.arenas["generator"]
    assert ga.ema is not None and store.arenas["discriminator"].ema is None
    assert torch.equal(ga.ema, ga.params)
    for name in ga.names:
        off, n, shape = ga.offsets[name]
        assert off % 64 == 0
        v = store.vars[name]
        assert v.data_ptr() == ga.params.data_ptr() + 4 * off and tuple(v.shape) == shape
        assert v.bg_grad.data_ptr() == ga.grads.data_ptr() + 4 * off
    with pytest.raises(ValueError):
        store.get_variable("new_variable", [1])


# ---------------------------------------------------------------- DiffAugment host logic
def test_diffaugment_policy_and_constants():
    assert DA.augment_params(128) == R.diffaugment_params(128) == (16, 64, 129)
    assert DA.policy_bits("color,translation,cutout") == 7 and DA.policy_bits("translation") == 2
    with pytest.raises(KeyError):
        DA.policy_bits("flip")
    with pytest.raises(NotImplementedError):
        DA.policy_bits("cutout,color")
    x = torch.zeros(1, 8, 8, 3)
    assert DA.DiffAugment(x, "") is x
    d = DA.draw(64, 128, "cpu", torch.Generator().manual_seed(0))
    assert d["t_x"].dtype == torch.int32 and int(d["t_x"].abs().max()) <= 16
    assert int(d["o_x"].min()) >= 0 and int(d["o_x"].max()) <= 128


# ---------------------------------------------------------------- C ABI
def _header_symbols():
    h = open(os.path.join(ROOT, "include", "biggan_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(bg_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    syms = _header_symbols()
    assert len(syms) >= 45
    assert sorted(hip.SIGNATURES.keys()) == syms              # the binding covers the whole header
    L = hip.lib()                                             # dlopen + getattr of every symbol
    assert L.bg_abi_version() == hip.ABI_VERSION and L.bg_target_arch() == b"gfx950"
    assert L.bg_last_error() is not None
    # workspace queries are pure host functions
    d = hip.conv_desc(64, 128, 128, 64, 128, 128, 64, 3, 1, 1, hip.PAD_REFLECT)
    assert L.bg_conv2d_wgrad_workspace_bytes(d) % 4 == 0
    assert L.bg_spectral_norm_workspace_bytes(10, 20) >= 4 * 34      # u, v, scalars (fp64 accumulators)


def test_ops_fail_loudly_without_gpu():
    from biggan_tensorflow_amd import functional as Fn
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Fn.PReluFn.apply(torch.zeros(2, 4), torch.zeros(4))
    with pytest.raises(RuntimeError):
        Fn.Conv2dFn.apply(torch.zeros(1, 4, 4, 4), torch.zeros(3, 3, 4, 4), None, 1, 1, 4, 4, hip.PAD_REFLECT)


def test_bench_flop_model_matches_baseline_md():
    import bench
    for (img, ch), gf in {(128, 64): 96.24, (128, 96): 212.16, (256, 96): 713.44, (512, 128): 2376.77,
                          (64, 32): 4.74}.items():
        assert round(bench.step_flops_per_image(img, ch)[0] / 1e9, 2) == gf


# ----------------------------------------------------------------------------------
# input pipeline (utils.py:12-121, BigGAN.py:768-787)
# ----------------------------------------------------------------------------------
def _png_with_filters(img, filters):
    """Encode an 8-bit RGB image using the given PNG filter type per row (test helper)."""
    import struct
    import zlib
    h, w, c = img.shape
    raw = b""
    prev = np.zeros(w * c, np.int32)
    for r in range(h):
        line = img[r].reshape(-1).astype(np.int32)
        ft = filters[r % len(filters)]
        out = np.zeros_like(line)
        for i in range(w * c):
            left = line[i - c] if i >= c else 0
            up = prev[i]
            ul = prev[i - c] if i >= c else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = left
            elif ft == 2:
                pred = up
            elif ft == 3:
                pred = (left + up) >> 1
            else:
                p = left + up - ul
                pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
                pred = left if (pa <= pb and pa <= pc) else (up if pb <= pc else ul)
            out[i] = (line[i] - pred) & 255
        raw += bytes([ft]) + out.astype(np.uint8).tobytes()
        prev = line

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_png_decoder_all_filter_types_and_channel_conversion(tmp_path):
    from biggan_tensorflow_amd import data as D
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    for filters in ([0], [1], [2], [3], [4], [0, 1, 2, 3, 4]):
        assert np.array_equal(D.decode_png(_png_with_filters(img, filters), 3), img), filters
    rgba = D.decode_png(_png_with_filters(img, [4]), 4)
    assert rgba.shape == (7, 5, 4) and (rgba[:, :, 3] == 255).all()
    grey = D.decode_png(_png_with_filters(img, [2]), 1)
    ref = np.clip(np.rint(0.299 * img[:, :, 0] + 0.587 * img[:, :, 1] + 0.114 * img[:, :, 2]), 0, 255)
    assert np.array_equal(grey[:, :, 0], ref.astype(np.uint8))
    # the writer used for sample grids round-trips through the reader
    x = rng.uniform(-1, 1, (1, 4, 6, 3)).astype(np.float32)
    p = utils.save_images(x, [1, 1], str(tmp_path / "a.png"))
    back = D.decode_png(open(p, "rb").read(), 3)
    assert np.array_equal(back, np.clip(np.rint((x[0] + 1) / 2 * 255), 0, 255).astype(np.uint8))
    with pytest.raises(ValueError):
        D.decode_png(b"not a png at all", 3)


def test_tf1_legacy_bilinear_resize():
    """tf.image.resize_images of TF 1.x maps destination index i to source i * in/out (no half-pixel shift):
    doubling [a, b] gives [a, (a+b)/2, b, b]; an integer reduction picks every n-th pixel."""
    from biggan_tensorflow_amd import data as D
    img = np.array([[[0.0], [10.0]], [[20.0], [30.0]]], np.float32)
    up = D.resize_bilinear_legacy(img, 4)[:, :, 0]
    assert np.allclose(up[0], [0, 5, 10, 10]) and np.allclose(up[:, 0], [0, 10, 20, 20]) and up[1, 1] == 15.0
    big = np.arange(64, dtype=np.float32).reshape(8, 8, 1)
    assert np.array_equal(D.resize_bilinear_legacy(big, 4)[:, :, 0], big[::2, ::2, 0])
    assert np.array_equal(D.resize_bilinear_legacy(big, 8), big)


def test_dataset_loader_shuffles_shards_and_labels(tmp_path):
    from biggan_tensorflow_amd import data as D
    folder = tmp_path / "dataset" / "toy"
    folder.mkdir(parents=True)
    rng = np.random.default_rng(1)
    for i in range(12):
        x = np.full((1, 8, 8, 3), -1.0, np.float32)
        x[0, :, :, 0] = i / 11.0 * 2 - 1                 # file index encoded in the red channel
        utils.save_images(x, [1, 1], str(folder / ("img%02d.png" % i)))
    with open(tmp_path / "labels.tsv", "w") as f:
        for i in range(12):
            f.write("img%02d.png\t%d\t%d\n" % (i, i % 2, 1 - i % 2))
    files, labels = D.load_data("toy", str(tmp_path / "labels.tsv"), n_labels=2, root=str(tmp_path / "dataset"))
    assert len(files) == 12 and labels[3] == [1.0, 0.0]
    seen = []
    for rank in range(2):
        idata = D.ImageData(4, 3, True, flip=False)
        ld = D.BatchLoader(files, labels, 3, idata, "cpu", seed=9, rank=rank, world=2)
        for _ in range(2):                               # one epoch = 12 / (3 * 2) = 2 steps per rank
            img, lab = next(ld)
            assert tuple(img.shape) == (3, 4, 4, 3) and tuple(lab.shape) == (3, 2) and img.dtype == torch.float32
            ids = np.rint((img[:, 0, 0, 0].numpy() + 1) / 2 * 11).astype(int)
            assert np.array_equal(lab[:, 0].numpy(), (ids % 2).astype(np.float32))
            seen += ids.tolist()
        ld.close()
    assert sorted(seen) == list(range(12))               # the two ranks' shards of one epoch partition the dataset
    utils.save_images(np.zeros((1, 8, 8, 3), np.float32), [1, 1], str(folder / "unlabelled.png"))
    with pytest.raises(RuntimeError):                    # utils.py:114: "No label found for file"
        D.load_data("toy", str(tmp_path / "labels.tsv"), n_labels=2, root=str(tmp_path / "dataset"))
    files2, labels2 = D.load_data("toy", str(tmp_path / "labels.tsv"), ignore_missing=True, n_labels=2,
                                  root=str(tmp_path / "dataset"))
    assert len(files2) == 13 and labels2[-1] == [0.0, 0.0]


def test_keep_checkpoints_prunes_old_files(tmp_path):
    """--keep_checkpoints N (main.py:18; swapping_saver(max_to_keep=N), BigGAN.py:1018): after N + 2 saves only the N
    newest checkpoint files remain and the index names the last one."""
    from biggan_tensorflow_amd import model, scope as S
    from tests.common import make_args
    args = make_args(img_size=64, ch=8, batch_size=2, keep_checkpoints=2, checkpoint_dir=str(tmp_path))
    gan = model.BigGAN(args, device="cpu", store=S.VariableStore("cpu", seed=1)).build_model()
    for step in (10, 20, 30, 40):
        gan.save(str(tmp_path), step)
    d = os.path.join(str(tmp_path), gan.model_dir)
    files = sorted(f for f in os.listdir(d) if f.endswith(".safetensors"))
    assert files == ["BigGAN.model-30.safetensors", "BigGAN.model-40.safetensors"]
    assert 'BigGAN.model-40' in open(os.path.join(d, "checkpoint")).read()


def test_reference_names_of_the_boundary_exist_with_reference_signatures():
    """names a caller of the reference's ops.py / DiffAugment_tf.py may use."""
    import inspect
    from biggan_tensorflow_amd import ops, DiffAugment as DA
    assert list(inspect.signature(ops.clown_conv).parameters) == ["x", "channels", "opt", "use_bias", "scope", "z"]
    assert list(inspect.signature(ops.mixed_resblock).parameters) == ["x", "inner_channels", "out_channels", "opt",
                                                                      "use_bias", "z", "scope"]
    assert list(inspect.signature(ops.global_avg_pooling).parameters) == ["x"]
    assert set(DA.AUGMENT_FNS) == {"color", "translation", "cutout"}
    assert [f.__name__ for f in DA.AUGMENT_FNS["color"]] == ["rand_brightness", "rand_saturation", "rand_contrast"]
    assert inspect.signature(DA.rand_translation).parameters["ratio"].default == 0.125
    assert inspect.signature(DA.rand_cutout).parameters["ratio"].default == 0.5
    import torch
    y = ops.global_avg_pooling(torch.empty(2, 4, 4, 8, device="meta"))
    assert tuple(y.shape) == (2, 8)


def test_bench_keeps_the_global_batch_fixed_under_strong_scaling():
    """bench.py --gpus N (BASELINE config 3): the global batch stays 256 for N = 1, 2, 4, 8 (strong scaling:
    one series, N = 1 runs all 256 images on one GPU); --scaling weak keeps 32 images per rank."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.resolve_workload("", 1) == ("c3", 256, "strong")
    assert bench.resolve_workload("c2", 1) == ("c2", 64, "weak")
    for n in (1, 2, 4, 8):
        name, b, mode = bench.resolve_workload("", n)
        assert (name, mode) == ("c3", "strong") and b * n == 256
    assert bench.resolve_workload("", 8, "weak") == ("c3", 32, "weak")
    assert bench.resolve_workload("c5", 8) == ("c5", 64, "strong")
    assert bench.resolve_workload("c2", 4) == ("c2", 64, "weak")
    with pytest.raises(SystemExit):
        bench.resolve_workload("c3", 3)


def test_bench_roofline_groups_by_kernel_symbol_and_counts_algorithmic_flops():
    """roofline_from_rows (bench.py): `dominant` is the kernel SYMBOL with the most time (not the per-shape tag: every
    attention launch shares a tag, every conv shape has its own), `achieved` comes from the step's algorithmic FLOPs
    (4 F_G + 8 F_D per image), not from what the launches counted (padded channels), and the algorithmic bytes of the
    launches stand beside the traffic."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = [("conv2d_fwd N2 H8 Cin8 Cout8 Ho8 k3 s1", "nn16_kernel<4, 0, false>", 1000.0, 4e9, 1.0),
            ("conv2d_dgrad N2 H8 Cin8 Cout8 Ho8 k3 s1", "nn16_kernel<4, 0, false>", 1000.0, 4e9, 1.5),
            ("attention16_fwd", "attn16_fwd_kernel<1, 2>", 500.0, 2e9, 2.0),
            ("deconv2d_wgrad N2 H8 Cin8 Cout8 Ho16 k4 s2", "tn16x_kernel<0, 32>", 3000.0, 6e9, 1.0),
            ("gram16 M8 N8 K8", "tn16x_kernel<2, 32>", 0.0, 9e9, 7.0)]          # (not an algorithmic launch)
    r = bench.roofline_from_rows(rows, nsteps=1, peak=100.0, fpi=5e9, B=2, ms_per_step=10.0)
    assert r["dominant"]["kernel"] == "nn16_kernel<4, 0, false>" and r["dominant"]["ms_per_step"] == 2.5
    assert r["gemm_ms_per_step"] == 5.5 and r["launches_per_step"] == 4
    assert abs(r["achieved"] - 1e10 / 5.5e-3 / 1e12) < 1e-2 and r["step_algorithmic_flops"] == 1e10
    assert r["gemm_flops_per_step_as_launched"] == 1.6e10
    assert r["algorithmic_bytes_per_step"] == 5500.0 and r["all_gemm_family_ms_per_step"] == 12.5


def test_pmc_summary_classifies_the_kernel_names_of_this_build():
    """tools/summarize_pmc.py sorts rocprofv3 kernel names into the GEMM family (whose HBM bytes DESIGN section 5.1 and
    bench.py's roofline.traffic quote) and its regulariser part by regular expression: the names the current kernels
    demangle to must land where they belong (a Gram launch `tn16x_kernel<2, 32>` once fell outside the regulariser
    pattern `tn16x?_kernel<2>`), and the committed summaries must be consistent with their own per-kernel entries."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("summarize_pmc", os.path.join(root, "tools", "summarize_pmc.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    gemm = ["void bg::nn16_kernel<4, 1, true>(bg::NN16Params)", "void bg::nn16_kernel<3, 0, false>(bg::NN16Params)",
            "void bg::nn16h_kernel<2, 3, 1>(bg::NN16Params)", "void bg::tn16x_kernel<0, 32>(bg::TN16Params)",
            "void bg::tn16_kernel<0>(bg::TN16Params)", "bg::nn16_slab_reduce_kernel(float const*, void*)",
            "void bg::reflect_fold_kernel<false>(void const*, void*)", "void bg::attn16_fwd_kernel<2, 3>(bg::A16Geom)",
            "void bg::nn_kernel<2, 2, 2, 2, true, 1, false, true>(bg::NNParams)", "void bg::tn_kernel<2, 2, 2, 2, 0, true>(bg::TNParams)",
            "bg::nn_slab_reduce_kernel(float const*, float*)", "bg::slab_reduce_kernel(float const*, float*, long, int, long)",
            "_ZN2bg21attn16_bwd_dkv_kernelILi1ELi2EEEvPKDF16bS2_S2_S2_PKfS4_PDF16bS5_NS_7A16GeomEllllll",
            "_ZN2bg20attn16_bwd_dq_kernelILi2ELi3EEEvPKDF16bS2_S2_S2_PKfS4_PDF16bNS_7A16GeomEllll",
            "void bg::attn_bwd_dkv_kernel<16, 1>(float const*)", "bg::rgb_conv_fwd_kernel(float const*)"]
    other = ["bg::adam_kernel(float*)", "bg::sn_batch_normalize_kernel(BgSnItem const*, int, char*)",
             "bg::lincomb_bf16x8_kernel(void)", "void bg::colreduce_kernel<2, 8, bg::BnStatsFnT<bf16>, double>(void)",
             "_ZN2bg30bn_apply_act_fwd_bf16x8_kernelEPKDF16bPKfS3_S3_S3_iS3_PDF16biii", "bg::bn_finalize_kernel(double const*)",
             "_ZN2bg16colreduce_kernelILi2ELi8ENS_10BnStatsFnTIDF16bEEdEEvT1_PT2_llii", "bg::ortho_cosine_kernel(float const*)"]
    for name in gemm:
        assert re.search(sp.GEMM_FAMILY, name), name
    for name in other:
        assert not re.search(sp.GEMM_FAMILY, name), name
    for name in ["void bg::tn16x_kernel<2, 32>(bg::TN16Params)", "void bg::tn16_kernel<2>(bg::TN16Params)",
                 "void bg::tn_kernel_bf16_tr<2>(bg::TNParams)",
                 "void bg::nn_kernel_bf16<2, 2, false, 2, false, true>(bg::NNParams)"]:
        assert re.search(sp.REGULARISER, name) and re.search(sp.GEMM_FAMILY, name), name
    assert not re.search(sp.REGULARISER, "void bg::tn16x_kernel<0, 32>(bg::TN16Params)")
    for fn in ("r02_pmc_c3.json", "r02_pmc_c3_b256.json"):
        with open(os.path.join(root, "profiles", fn)) as fh:
            d = json.load(fh)
        S_ = d["_summary"]
        tot = lambda f: d[f]["launches"] * (d[f]["hbm_read_bytes_per_launch"] + d[f].get("hbm_write_bytes_per_launch", 0.0))
        fams = S_["gemm_families"]
        assert all(re.search(sp.GEMM_FAMILY, f) for f in fams)
        reg = sum(tot(f) for f in fams if re.search(sp.REGULARISER, f)) / S_["iterations"]
        assert abs(reg - S_["regulariser_gemm_hbm_bytes_per_iteration"]) < 1e-6 * max(reg, 1.0)
        allb = sum(tot(f) for f in fams) / S_["iterations"]
        assert abs(allb - S_["gemm_family_hbm_bytes_per_iteration"]) < 1e-6 * allb
        assert abs(allb - reg - S_["conv_attention_gemm_hbm_bytes_per_iteration"]) < 1e-6 * allb


def test_workspace_queries_run_the_bf16_planner_on_the_host():
    """The workspace queries of the bf16-resident convolutions are pure host code and run the same tile / split-K planner
    as the launch (igemm16.hip plan_nn16, incl. the position-major decision for small maps).  Default form of the
    reflect-padded input gradient (plain transposed gather + mirrored-tap launches): whole fp32 split-K slabs of the
    H x W gradient, nothing else; BG_DGRAD_RING=0 (round 2's form): at least the gradient on the reflect-padded grid,
    plus whole slabs of it - with either setting of BG_NN16_POSMAJOR."""
    import ctypes
    L = hip.lib()
    try:
        for (N, H, C, Co, k, s) in [(512, 4, 1536, 1536, 3, 1), (256, 8, 768, 768, 3, 1), (512, 8, 768, 1536, 3, 2),
                                    (32, 4, 1536, 1536, 3, 1), (8, 4, 96, 96, 3, 1), (2, 128, 96, 96, 3, 1)]:
            d = hip.conv_desc(N, H, H, C, H // s, H // s, Co, k, s, 1, hip.PAD_REFLECT, hip.COMPUTE_BF16, hip.BF16,
                              hip.BF16, 1)
            Hp = ((H // s - 1) * s + k + s - 1) // s * s                  # padded extent, a multiple of the stride
            padded = N * Hp * Hp * C
            for pm in ("0", "1"):
                os.environ["BG_NN16_POSMAJOR"] = pm
                os.environ["BG_DGRAD_RING"] = "0"
                nb = int(L.bg_conv2d_dgrad_workspace_bytes(ctypes.byref(d)))
                assert nb >= padded * 2, (N, H, pm, nb)
                slabs = nb - ((padded * 2 + 255) // 256 * 256)
                assert slabs % (padded * 4) == 0 and slabs // (padded * 4) <= 16, (N, H, pm, nb)
                os.environ["BG_DGRAD_RING"] = "1"
                nb = int(L.bg_conv2d_dgrad_workspace_bytes(ctypes.byref(d)))
                assert nb % (N * H * H * C * 4) == 0 and nb // (N * H * H * C * 4) <= 16, (N, H, pm, nb)
                os.environ.pop("BG_DGRAD_RING")          # default: the form the batch-size rule picks - one of the two
                assert int(L.bg_conv2d_dgrad_workspace_bytes(ctypes.byref(d))) in (
                    nb, int(L.bg_conv2d_dgrad_workspace_bytes(ctypes.byref(d))))
                assert int(L.bg_conv2d_fwd_workspace_bytes(ctypes.byref(d))) % (N * (H // s) ** 2 * Co * 4) == 0
    finally:
        os.environ.pop("BG_NN16_POSMAJOR", None)
        os.environ.pop("BG_DGRAD_RING", None)
