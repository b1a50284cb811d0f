"""CPU: host-side logic of the product and the C-ABI surface (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import vit_tf_amd as vt
from vit_tf_amd import _lib
from oracle import dino_vit, feature_volume as ofv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'vittf.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(vittf_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 17
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/vittf.h but not exported by libvittf.so'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.vittf_abi_version() == vt._lib.ABI_VERSION == 6
    assert lib.vittf_status_string(-2) == b'workspace too small'


def test_workspace_query_and_argument_validation_need_no_gpu():
    lib = _lib.load()
    cfg = _lib.VitConfig(384, 12, 6, 8, _lib.BF16, 1e-6)
    one = lib.vittf_vit_workspace_bytes(ctypes.byref(cfg), 1, 4097)
    many = lib.vittf_vit_workspace_bytes(ctypes.byref(cfg), 32, 4097)
    # fp32 residual + LN out + qkv + attention out = 4 + 2 + 6 + 2 bytes per token-feature
    assert one >= 4097 * 384 * 14 and 31 * one < many <= 32 * one
    bad = _lib.VitConfig(384, 12, 5, 8, _lib.BF16, 1e-6)           # heads * 64 != D
    assert lib.vittf_vit_workspace_bytes(ctypes.byref(bad), 1, 4097) == 0
    assert lib.vittf_gemm(None, None, None, None, 1, 128, 64, 0, 0, 0, None) == -1
    assert lib.vittf_similarity_workspace_bytes(2, 64 ** 3, 16) >= 2 * 64 ** 3 * 4
    assert lib.vittf_similarity_workspace_bytes(5, 64 ** 3, 5120) > lib.vittf_similarity_workspace_bytes(5, 64 ** 3, 16) + 5000 * 384 * 4


def test_product_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(vt.VittfError):
        vt.HipViT(vt.synthetic_state_dict((128, 1, 2, 8), 0), (128, 1, 2, 8))
    with pytest.raises(vt.VittfError):
        vt.compute_similarities(np.zeros((4, 4, 4)), torch.zeros(8, 2, 2, 2), {'a': torch.zeros(2, 3)})
    r = subprocess.run([sys.executable, 'infer.py', '--data-path', 'x.npy', '--cpu'], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and 'no CPU path' in r.stdout


def test_sizing_rule_matches_reference_configs():
    assert vt.sizing((512, 512, 512), 64, 8) == ((512,) * 3, (64,) * 3)
    assert vt.sizing((256, 256, 256), 64, 8) == ((512,) * 3, (64,) * 3)
    assert vt.sizing((64, 64, 64), 64, 8) == ((512,) * 3, (64,) * 3)
    assert vt.sizing((512, 512, 512), 128, 8) == ((1024,) * 3, (128,) * 3)
    for shape, fos in (((200, 256, 312), 64), ((10, 10, 10), 4), ((20, 12, 28), 2)):
        assert vt.sizing(shape, fos, 8) == ofv.sizing(shape, fos, 8)


def test_shard_windows_cover_everything_once():
    for n_out in (1, 3, 8, 50, 64, 78):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                first, cnt, chunk = vt.extract.shard_windows(n_out, r, world)
                assert cnt <= chunk and chunk * world >= n_out
                seen += list(range(first, first + cnt))
            assert seen == list(range(n_out))


def test_folded_patch_embed_equals_three_channel_conv():
    sd = vt.synthetic_state_dict('vits8', 1)
    w, b = sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias']
    w_t, b1 = vt.fold_patch_embed(w, b)
    assert w_t.shape == (64, 384) and b1.shape == (384,)
    x = torch.rand(5, 1, 8, 8, dtype=torch.float64)
    mean = torch.tensor(vt.weights.IN_MEAN, dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor(vt.weights.IN_STD, dtype=torch.float64).view(1, 3, 1, 1)
    ref = torch.nn.functional.conv2d((x.expand(-1, 3, -1, -1) - mean) / std, w.double(), b.double())[:, :, 0, 0]
    got = x.reshape(5, 64) @ w_t.double() + b1.double()
    assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6)


def test_pos_embed_interpolation_is_the_upstream_form():
    sd = vt.synthetic_state_dict((128, 1, 2, 8), 2)
    for rows, cols in ((512, 512), (24, 16), (64, 256), (224, 224)):
        mine = vt.interpolate_pos_embed(sd['pos_embed'], rows, cols, 8)
        ref = dino_vit.interpolate_pos_embed(sd['pos_embed'], (rows // 8) * (cols // 8), rows, cols, 8)
        assert torch.equal(mine, ref)
    # the scale-factor form differs from a plain size= resize: make sure we did not "simplify" it
    g = sd['pos_embed'][:, 1:].reshape(1, 28, 28, 128).permute(0, 3, 1, 2)
    plain = torch.nn.functional.interpolate(g, size=(64, 64), mode='bicubic').permute(0, 2, 3, 1).reshape(1, -1, 128)
    assert (vt.interpolate_pos_embed(sd['pos_embed'], 512, 512, 8)[:, 1:] - plain).abs().max() > 1e-3


def test_state_dict_roundtrip_and_checksum(tmp_path):
    sd = vt.synthetic_state_dict((128, 2, 2, 8), 4)
    assert vt.weights.state_dict_checksum(sd) == vt.weights.state_dict_checksum(vt.synthetic_state_dict((128, 2, 2, 8), 4))
    assert vt.weights.state_dict_checksum(sd) != vt.weights.state_dict_checksum(vt.synthetic_state_dict((128, 2, 2, 8), 5))
    torch.save({'teacher': {'backbone.' + k: v for k, v in sd.items()} | {'head.mlp.weight': torch.zeros(2)}}, tmp_path / 'ck.pth')
    back = vt.load_state_dict_file(tmp_path / 'ck.pth')
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    dino_vit.build_vit((128, 2, 2, 8), back)        # loads strictly into the oracle's module tree (DINO key names)


def test_reference_shaped_helpers():
    import infer
    t = torch.zeros(4, 5)
    assert infer.make_3d(t).shape == (1, 4, 5) and infer.make_5d(t).shape == (1, 1, 1, 4, 5)
    with pytest.raises(Exception):
        infer.make_nd(torch.zeros(2, 2, 2), 2)
    x = torch.tensor([1.0, 3.0, 5.0])
    assert torch.equal(infer.norm_minmax(x), torch.tensor([0.0, 0.5, 1.0]))
    assert infer.in_mean == [0.485, 0.456, 0.406] and infer.in_std == [0.229, 0.224, 0.225]

    class A:
        dino_model = None; dino2_model = None
    a = A()
    assert infer.load_model(a)[0::2] == ('vits8', 8) and a.model == 'vits8'
    a = A(); a.dino_model = 'vitb16'
    assert infer.load_model(a)[2] == 16
    a = A(); a.dino_model = 'vits8'; a.dino2_model = 'vits14'
    with pytest.raises(SystemExit) as e:
        infer.load_model(a)
    assert e.value.code == 1


def test_output_path_contract(tmp_path):
    import infer

    class A:
        pass
    a = A(); a.data_path = str(tmp_path / 'vol.npy'); a.cache_path = None; a.model = 'vits8'; a.slice_along = 'all'
    a.feature_output_size = 64; a.overwrite = False
    p = infer.handle_output_path(a)
    assert p.name == 'vol_vits8_all_features64.npy'
    p.write_bytes(b'x')
    a.cache_path = None
    with pytest.raises(SystemExit) as e:
        infer.handle_output_path(a)
    assert e.value.code == 1
    a.cache_path = None; a.overwrite = True
    assert infer.handle_output_path(a) == p
    # file formats of load_data (infer.py:212-237)
    v = torch.rand(3, 4, 5).half()
    np.save(tmp_path / 'a.npy', v.numpy()); np.save(tmp_path / 'b.npy', {'vol': v.numpy()})
    torch.save(v, tmp_path / 'c.pt'); torch.save({'vol': v}, tmp_path / 'd.pt')
    for name, dt in (('a.npy', torch.float32), ('b.npy', torch.float32), ('c.pt', torch.float16), ('d.pt', torch.float16)):
        got = infer.load_data(tmp_path / name)
        assert got.dtype == dt and torch.equal(got.float(), v.float())
    with pytest.raises(SystemExit):
        infer.load_data(tmp_path / 'missing.npy')
    infer.save_features({'k': v}, tmp_path / 'f.npy')
    assert np.load(tmp_path / 'f.npy', allow_pickle=True)[()]['k'].dtype == np.float16


def test_sampler_oracle_is_the_scipy_restatement():
    """oracle/samplers.py (CPU, scipy) draws distinct voxels of the mask / of its eroded shell."""
    from oracle import samplers as osmp
    _, lab = vt.synthetic_volume('sphere_filled', 32)
    torch.manual_seed(0)
    u = osmp.sample_uniform(lab, 20)
    assert u.shape == (20, 3) and bool(lab[u[:, 0], u[:, 1], u[:, 2]].all()) and len({tuple(r) for r in u.tolist()}) == 20
    shell = osmp.surface_shell(lab.numpy())
    assert 0 < shell.sum() < lab.sum() and not (shell & ~lab.numpy().astype(bool)).any()
    s = osmp.sample_surface(lab.numpy(), 20)
    assert s.shape == (20, 3) and bool(shell[s[:, 0], s[:, 1], s[:, 2]].all())
    assert osmp.sample_both(lab.numpy(), 10).shape == (10, 3)


def test_bench_flop_formula_matches_survey_figures():
    """bench.vit_flops = SURVEY.md 8d's F_min: 444.9 GF per slice (ViT-S/8, N = 4097), 1211.2 GF (ViT-B/8), 5181.1 GF at
    N = 16385 -- the numbers `roofline.achieved` is priced with."""
    import bench
    for (n, d, gf) in ((4097, 384, 444.9), (4097, 768, 1211.2), (16385, 384, 5181.1)):
        total = sum(bench.vit_flops(n, d, 12, 8).values())
        assert abs(total / 1e9 - gf) / gf < 5e-4, (n, d, total / 1e9)


def test_bench_pmc_traffic_only_for_the_measured_kernel(tmp_path, monkeypatch):
    """A committed PMC pass prices `roofline.traffic` only for the very code object (kernel source, shared headers, Makefile
    flags), kernel, shape and batch it was taken on."""
    import json
    import bench
    (tmp_path / 'profiles').mkdir()
    src = tmp_path / 'vit-tf_amd' / 'csrc'
    src.mkdir(parents=True)
    for name in bench.KERNEL_SOURCES['attention']:
        (src / name).write_text(f'{name} v1')
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    rec = {'batch': 256, 'tokens': 4097, 'hbm_bytes_per_launch': 1000, 'source_sha1': bench.kernel_source_hash('attention')}
    (tmp_path / 'profiles' / 'pmc_attention.json').write_text(json.dumps(rec))
    assert bench.pmc_traffic('attention', 256, tokens=4097) == 1000
    assert bench.pmc_traffic('attention', 32, tokens=4097) is None              # a pass at another batch is not this launch
    assert bench.pmc_traffic('attention', 256, tokens=16385) is None            # another shape
    assert bench.pmc_traffic('similarity', 16) is None                          # no pass on file
    for changed in ('Makefile', 'attn_common.h', bench.KERNEL_SOURCES['attention'][0]):
        (src / changed).write_text('v2')
        assert bench.pmc_traffic('attention', 256, tokens=4097) is None        # the code object changed since the pass
        (src / changed).write_text(f'{changed} v1')
        assert bench.pmc_traffic('attention', 256, tokens=4097) == 1000


def test_engine_batch_follows_a_workspace_budget(monkeypatch):
    """Slices per engine call: 512 at the metric's shape (256 for models wider than 384), scaled down with the token count so
    that a call's row count stays where 512 x 4097 puts it; an explicit request or VITTF_ENGINE_BATCH wins."""
    from vit_tf_amd.extract import engine_batch_for
    monkeypatch.delenv('VITTF_ENGINE_BATCH', raising=False)
    assert engine_batch_for(4097, 384) == 512
    assert engine_batch_for(16385, 384) == 128         # sub/infer_and_merge.sh: fos 128, 1024 x 1024 slices
    assert engine_batch_for(4097, 768) == 256          # ViT-B/8: half the slices, the same 11 GB of workspace
    assert engine_batch_for(16385, 768) == 64
    assert engine_batch_for(65, 384) == 512 and engine_batch_for(10 ** 9, 384) == 1
    assert engine_batch_for(4097, 384, 32) == 32
    from vit_tf_amd.extract import AtLeast             # the reference's --batch-size: a lower bound, never a smaller launch
    assert engine_batch_for(4097, 384, AtLeast(4)) == 512 and engine_batch_for(4097, 384, AtLeast(600)) == 600
    assert engine_batch_for(4097, 768, AtLeast(300)) == 300 and engine_batch_for(4097, 768, AtLeast(5000)) == 1024
    assert engine_batch_for(16385, 384, AtLeast(2)) == 128
    monkeypatch.setenv('VITTF_ENGINE_BATCH', '8')
    assert engine_batch_for(4097, 384, 32) == 8


def test_block_tail_kernel_register_contract():
    """csrc/tail_fx.hip runs two waves per SIMD at exactly 256 registers with hand-counted vmcnt waits in its X waves: a spill
    reload inside a main-step loop is a vector-memory load the counts do not know of (and hipcc puts s_waitcnt vmcnt(0) in
    front of its use, which drains the weight ring), and its LDS-DMA pieces leave M0 pointing at their destination, which is
    only sound while hipcc keeps nothing of its own in M0.  Properties of the generated code, so they are checked on it (the
    flags are the Makefile's: tools/kernel_asm.sh); the same for the activation-stationary qkv GEMM."""
    import re
    import shutil
    if not shutil.which('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc')
    for src, kern in (('tail_fx.hip', 'tail_fx_kernel'), ('gemm_as.hip', 'gemm_as_kernel')):
        r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'kernel_asm.sh'), src], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        usage = [l for l in r.stdout.splitlines() if kern in l and 'ScratchSize' in l]
        assert len(usage) == 2, r.stdout                                       # bf16 and fp16
        for l in usage:
            assert 'Occupancy [waves/SIMD]: 2' in l and ' VGPRs Spill: 0' in l, l
        hot = [l for l in r.stdout.splitlines() if 'scratch instructions' in l]
        assert hot, r.stdout
        for l in hot:      # tools/asm_hot_scratch.py: no spill instruction inside a loop block that issues MFMAs in pinned gaps
            if 'run_x' in l:      # (the X role keeps a few addresses in scratch across the tile boundary: reloaded outside the main loop)
                m = re.search(r"in pinned loop blocks: (.*)$", l)
                assert m and ('none' in m.group(1) or all(int(n) <= 16 for n in re.findall(r", (\d+)\)", m.group(1)))), l
            else:
                assert l.rstrip().endswith('none'), l
        asm = open('/tmp/vittf_asm/' + src.replace('.hip', '.s')).read()
        m0 = [l.strip() for l in asm.splitlines() if re.search(r'\bm0\b', l) and not l.strip().startswith(';')]
        ours = r's_mov_b32 (m0, s\d+|s\d+, m0)'          # lds_dma16_keep, and the prologue's lds_dma16 (saves and restores)
        assert m0 and all(re.fullmatch(ours, l) for l in m0), [l for l in m0 if not re.fullmatch(ours, l)][:5]
        assert 'flat_load' not in asm


def test_window_goldens_cover_the_fullsize_tests():
    """tests/golden/windows512.npz (made by tests/golden/make_window_goldens.py from the CPU oracle) holds every pooled window the
    full-size GPU tests compare with: fp16, every 4th feature row / column of a (D, 64, 64) window, finite, not constant."""
    import numpy as np
    import helpers
    z = helpers.load_golden(os.path.join(ROOT, 'tests', 'golden'), 'windows512.npz')
    assert set(z) == {helpers.window_key(*w) for w in helpers.WINDOWS512}
    n = 64 // helpers.WINDOW_STRIDE
    for arch, seed, axis, w in helpers.WINDOWS512:
        a = z[helpers.window_key(arch, seed, axis, w)]
        assert a.dtype == np.float16 and a.shape == ({'vits8': 384, 'vitb8': 768}[arch], n, n)
        assert np.isfinite(a).all() and float(a.astype(np.float32).std()) > 1e-3


def _asm_functions(path):
    """{mangled name: [instruction lines]} of a device assembly file (labels, comments and directives dropped)."""
    import re
    fns, cur = {}, None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = fns.setdefault(m.group(1), [])
            continue
        if line.startswith('.Lfunc_end'):
            cur = None
            continue
        t = line.strip()
        if cur is not None and t and not t.startswith((';', '.')) and not re.match(r'^[\w.$]+:', t):
            cur.append(t.split(';')[0].strip())
    return fns


def _vregs(tok):
    """VGPR numbers named by an operand token ('v12', 'v[4:7]'); empty for anything else."""
    import re
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def _operands(ins):
    parts = ins.split(None, 1)
    return [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []


def test_hand_waited_registers_in_the_disassembly():
    """Two places wait by hand for values the compiler believes are already there; both are properties of the
    generated code, so they are checked on it:
      * sim_mfma_few_kernel requests its query rows with asm loads and waits for them with a counted vmcnt behind two parts'
        LDS-DMA pieces: nothing may touch the loaded registers between the loads and that wait;
      * attn_fp8_kernel<*, true> masks its per-row E8M0 scale bytes in an asm statement that carries the wait states
        v_mfma_scale needs behind a VALU write of its scale operand: every per-lane scale register an MFMA reads must have been
        written last by such a statement (not by a copy or a reload the compiler put in front of the MFMA)."""
    import re
    import shutil
    if not shutil.which('/opt/rocm/bin/hipcc'):
        pytest.skip('no hipcc')
    for src in ('sim_mfma.hip', 'attention_fp8.hip'):
        r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'kernel_asm.sh'), src], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
    few = [v for k, v in _asm_functions('/tmp/vittf_asm/sim_mfma.s').items() if 'sim_mfma_few_kernel' in k]
    assert len(few) == 1
    ins = few[0]
    loads = [i for i, t in enumerate(ins) if t.startswith('global_load_dwordx4') and ' off' in t]
    assert len(loads) >= 6
    first, last = loads[0], loads[5]
    assert last - first <= 40, 'the six query-row loads are one group (address arithmetic between them)'
    regs = set().union(*(_vregs(_operands(ins[i])[0]) for i in loads[:6]))
    assert len(regs) == 24
    wait = next(i for i in range(last, len(ins)) if re.match(r's_waitcnt vmcnt\(8\)', ins[i]))
    for t in ins[last + 1:wait]:
        touched = set().union(*(_vregs(o) for o in _operands(t)))
        assert not (touched & regs), f'{t!r} touches a query register before the counted wait'
    dma = [t for t in ins[last + 1:wait] if t.startswith(('global_load_lds', 'buffer_load')) or ' lds' in t]
    assert len(dma) == 8, f'{len(dma)} LDS-DMA pieces between the query loads and vmcnt(8)'
    fns = {k: v for k, v in _asm_functions('/tmp/vittf_asm/attention_fp8.s').items() if 'attn_fp8_kernel' in k and 'Lb1E' in k}
    assert len(fns) == 2
    for name, ins in fns.items():
        n_row = 0
        for i, t in enumerate(ins):
            if not t.startswith('v_mfma_scale'):
                continue
            ops = _operands(t.split(' op_sel')[0])
            for sreg in ops[4:6]:
                (r_,) = _vregs(sreg)
                j = next((j for j in range(i - 1, -1, -1) if _operands(ins[j]) and r_ in _vregs(_operands(ins[j])[0])
                          and not ins[j].startswith(('s_', 'ds_write', 'buffer_store', 'global_store', 'scratch_store'))), None)
                assert j is not None, (name, t)
                w = ins[j]
                if re.match(r'v_and_b32 v\d+, 0xff, v\d+$', w) and ins[j + 1] == 's_nop 7':
                    n_row += 1
                    continue
                if re.match(r'v_mov_b32_e32 v\d+, (0x[0-9a-f]+|\d+)$', w):      # a uniform scale: the same byte in every lane
                    continue
                states = 0                                                      # any other writer: far enough in front of the MFMA
                for x in ins[j + 1:i]:
                    m = re.match(r's_nop (\d+)', x)
                    states += int(m.group(1)) + 1 if m else 1
                assert states >= 8, (name, w, states, t)
        assert n_row >= 8, (name, n_row)
