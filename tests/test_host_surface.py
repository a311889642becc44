"""CPU tests of the host mirror: reference-compatible state_dict surface, the C ABI library loads and exports every
symbol of include/vec2wav_hip.h, and the product refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from tests.golden_util import load_golden
from wavthruvec_pytorch_amd import Generator, ConditionalBatchNorm1d, synthetic, _hip, workmodel, utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('name', ['rb2_train_b2_t8', 'rb1_train_b2_t8', 'rb2_1024_x640_train_b2_t8'])
def test_state_dict_keys_and_shapes_match_reference(name):
    z, meta = load_golden(name)
    h = synthetic.make_hparams(**meta['hp'])
    g = Generator(h)
    sd = g.state_dict()
    assert list(sd.keys()) == [str(k) for k in z['meta_keys']]
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in z['meta_shapes']]
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))          # strict load of a reference-format dict


def test_remove_weight_norm_renames_keys_like_reference():
    z, meta = load_golden('rb2_rmwn_train_b2_t8')
    h = synthetic.make_hparams(**meta['hp'])
    g = Generator(h)
    g.remove_weight_norm()
    assert list(g.state_dict().keys()) == [str(k) for k in z['keys_after_rmwn']]
    with pytest.raises(ValueError):
        g.conv_pre.remove_weight_norm()                                 # already removed, as torch raises


def test_default_hparams_select_resblock2():
    from wavthruvec_pytorch_amd import hparams as hp, ResBlock1, ResBlock2
    assert hp.resblock == 1 and hp.resblock != '1'
    g = Generator(synthetic.make_hparams(num_wv_feat=768))
    assert all(isinstance(r, ResBlock2) for r in g.resblocks) and len(g.resblocks) == 15
    g1 = Generator(synthetic.make_hparams(num_wv_feat=768, resblock='1'))
    assert all(isinstance(r, ResBlock1) for r in g1.resblocks)
    assert sum(p.numel() for p in g.parameters()) == 8581602           # SURVEY.md Q1 (oracle-verified count)
    assert sum(p.numel() for p in g1.parameters()) == 15926370


def test_no_cpu_fallback():
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h)
    x, spk, nz = synthetic.make_inputs(h, 1, 4)
    with pytest.raises(TypeError):
        g(x)
    with pytest.raises(RuntimeError, match='HIP path'):
        g(x, spk, nz)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ConditionalBatchNorm1d(16)(torch.randn(2, 16, 8), torch.randn(2, 128))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'vec2wav_hip.h')).read()
    declared = set(re.findall(r'\b(v2w_[a-z0-9_]+)\s*\(', header))
    declared -= {n for n in declared if n.endswith('_args')}
    assert declared, 'no prototypes parsed'
    assert declared == set(_hip.SIGNATURES), (declared ^ set(_hip.SIGNATURES))
    lib = ctypes.CDLL(_hip.lib_path())
    for name in declared:
        assert hasattr(lib, name), name
    loaded = _hip.load()
    assert loaded.v2w_abi_version() == _hip.ABI_VERSION
    assert loaded.v2w_build_arch() == b'gfx950'


def _header_struct_fields(header, name):
    """Field names of `typedef struct { ... } name;` in declaration order (comments stripped, array suffixes dropped)."""
    end = header.index('} %s;' % name)
    body = header[header.rindex('typedef struct {', 0, end):end]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    groups = re.findall(r'\b(?:const\s+)?(?:float|double|int32_t|int64_t|void)\s*\*?\s*([a-zA-Z_0-9\[\]]+(?:\s*,\s*[a-zA-Z_0-9\[\]]+)*)\s*;', body)
    return [re.sub(r'\[.*?\]', '', n).strip() for grp in groups for n in grp.split(',')]


@pytest.mark.parametrize('cname,mirror', [('v2w_conv1d_args', 'Conv1dArgs'), ('v2w_convt1d_args', 'ConvT1dArgs'),
                                          ('v2w_stage_split_args', 'StageSplitArgs'), ('v2w_stage_args', 'StageArgs')])
def test_struct_layouts_match_header_field_order(cname, mirror):
    """The ctypes mirrors list the header's fields in the header's order (ABI v28 added splitk_ws / splitk_ws_bytes to both conv structs)."""
    header = open(os.path.join(ROOT, 'include', 'vec2wav_hip.h')).read()
    flat = _header_struct_fields(header, cname)
    want = [f[0].rstrip('_') if f[0] != '_pad' else '_pad' for f in getattr(_hip, mirror)._fields_]
    assert flat == want, (flat, want)


def test_library_keeps_no_allocations_or_process_state():
    """the C ABI allocates nothing and keeps no mutable state - workspaces are the caller's (v2w_conv1d_args::splitk_ws,
    the slab arguments of v2w_wgrad*).  Checked on the sources: no hipMalloc / hipFree and no function-local or file-level mutable static."""
    csrc = os.path.join(ROOT, 'wavthruvec_pytorch_amd', 'csrc')
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith(('.hip', '.h')):
            continue
        text = re.sub(r'//.*', '', open(os.path.join(csrc, fn)).read())
        assert not re.search(r'\bhip(Malloc|Free|MallocAsync|HostMalloc)\b', text), fn
        for m in re.finditer(r'^\s*static\s+(?!inline|constexpr|const\b|__device__|int\s+\w+\s*\(|bool\s+\w+\s*\(|\w+\s+\w+\s*\()([^;(]*);', text, flags=re.M):
            if 'V2W_TIMELINE' in fn or 'g_tl' in m.group(0):      # diagnostic builds only (-DV2W_TIMELINE)
                continue
            raise AssertionError(f'{fn}: mutable static `{m.group(0).strip()}`')


def test_workmodel_matches_survey_contract_figures():
    h = synthetic.make_hparams(num_wv_feat=768)
    f, b = workmodel.totals(h, 32, 256)
    n = 32 * 256 * 320
    assert abs(f / n - 371526.4) < 1 and abs(b / n - 3878.9) < 1       # 371.5 kFLOP, 3 879 B per sample
    assert abs(f / 1e9 - 973.9) < 0.1 and abs(b / 1e9 - 10.17) < 0.01


def test_checkpoint_helpers(tmp_path):
    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h)
    p = os.path.join(str(tmp_path), 'g_00005000')
    utils.save_checkpoint(p, {'generator': g.state_dict()})            # train.py:228-230 format
    utils.save_checkpoint(os.path.join(str(tmp_path), 'g_00000100'), {'generator': g.state_dict()})
    assert utils.scan_checkpoint(str(tmp_path), 'g_') == p
    sd = utils.load_checkpoint(p, 'cpu')['generator']
    Generator(h).load_state_dict(sd)
    assert utils.get_padding(11, 3) == 15 and utils.get_padding(7) == 3


def test_synthesize_wire_formats(tmp_path):
    """text2vec `.npy` (1,T,C) -> (1,C,T); `{spk}.pth` (1,1,192) -> (1,192); 16-bit PCM wav writer."""
    import wave
    import numpy as np
    from wavthruvec_pytorch_amd import synthesize as S
    a = np.random.default_rng(0).standard_normal((1, 13, 768)).astype(np.float32)
    np.save(os.path.join(str(tmp_path), 'u_feat_postnet.npy'), a)
    x = S.load_latents(os.path.join(str(tmp_path), 'u_feat_postnet.npy'))
    assert x.shape == (1, 768, 13) and x.is_contiguous()
    assert torch.equal(x[0, :, 3], torch.from_numpy(a[0, 3]))
    torch.save(torch.randn(1, 1, 192), os.path.join(str(tmp_path), 'SSB0005.pth'))
    assert S.load_speaker_embedding(os.path.join(str(tmp_path), 'SSB0005.pth')).shape == (1, 192)
    y = torch.tensor([[[0.0, 0.5, -1.0, 1.0, 2.0]]])
    S.write_wav(os.path.join(str(tmp_path), 'o.wav'), y, 16000)
    with wave.open(os.path.join(str(tmp_path), 'o.wav')) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 16000, 5)
        pcm = np.frombuffer(w.readframes(5), dtype='<i2')
    assert list(pcm) == [0, 16384, -32767, 32767, 32767]
    with pytest.raises(ValueError):
        np.save(os.path.join(str(tmp_path), 'bad.npy'), np.zeros((2, 3, 4), np.float32))
        S.load_latents(os.path.join(str(tmp_path), 'bad.npy'))


def test_discriminator_state_dict_surface():
    """MultiPeriodDiscriminator(hp) / MultiScaleDiscriminator() hold the reference's keys and shapes (models.py:158-258; the spec
    itself is asserted against the reference modules by tools/gen_disc_goldens.py), and refuse to run off the GPU."""
    import pytest
    import torch
    from types import SimpleNamespace
    from wavthruvec_pytorch_amd import synthetic
    from wavthruvec_pytorch_amd.discriminators import MultiPeriodDiscriminator, MultiScaleDiscriminator
    mpd = MultiPeriodDiscriminator(SimpleNamespace(periods=[13, 17, 19]))
    msd = MultiScaleDiscriminator()
    for m, spec in ((mpd, synthetic.mpd_state_dict_spec()), (msd, synthetic.msd_state_dict_spec())):
        sd = m.state_dict()
        assert list(sd.keys()) == [k for k, _, _ in spec]
        for k, shape, _ in spec:
            assert tuple(sd[k].shape) == shape, k
        m.load_state_dict(synthetic.make_disc_state_dict(spec, seed=1))
    y = torch.zeros(1, 1, 1000)
    with torch.no_grad(), pytest.raises(RuntimeError):
        mpd(y, y)
    with pytest.raises(RuntimeError):              # under autograd too: CPU tensors are refused, nothing falls back to torch
        msd(y, y)
    # set_precision: 'f32' / 'f16x3' on every conv of the module, anything else refused; the state_dict surface is untouched
    from wavthruvec_pytorch_amd.discriminators import set_precision, _DiscConv
    assert set_precision(mpd, 'f16x3') is mpd and all(l.precision == 'f16x3' for l in mpd.modules() if isinstance(l, _DiscConv))
    assert list(mpd.state_dict().keys()) == [k for k, _, _ in synthetic.mpd_state_dict_spec()]
    with pytest.raises(ValueError):
        set_precision(msd, 'bf16')


def test_bench_refuses_a_run_smaller_than_the_one_asked_for():
    """bench.py --gpus N without a launcher either starts N ranks or fails: with no GPU visible it must exit non-zero BEFORE
    touching a device and print no JSON line (it used to print n_gpus 1 for a --gpus 8 request)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    env['HIP_VISIBLE_DEVICES'] = ''
    env['CUDA_VISIBLE_DEVICES'] = ''
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8'], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode != 0
    assert 'only 0 GPU' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
    # a launcher that provides a different world size than --gpus is an error too
    env2 = dict(env, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8'], capture_output=True, text=True, env=env2,
                        timeout=300)
    assert r2.returncode != 0 and 'WORLD_SIZE=2' in r2.stderr


def test_bench_line_is_compact_and_ends_in_the_summary():
    """The driver keeps the last ~8 KB of the bench line: the compact form of a full round-4 line (every block present) must fit, must keep
    the contract's roofline / cpu_baseline keys, and must END in `summary` with one [ms, hbm_frac, mfma_frac] entry per block."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    with open(os.path.join(ROOT, 'profiles', 'r04_cfg2_bench.json')) as f:
        out = json.load(f)
    blocks = ('alt_precision', 'cfg3_bf16', 'cfg2_bf16', 'cfg5_f32', 'resblock1_f32', 'resblock1_bf16', 'train_step', 'stat_sync')
    out['roofline'] = b.compact_roofline(out['roofline'])
    out['cpu_baseline'] = b.compact_cpu(out['cpu_baseline'])
    for n in blocks:
        out[n] = b.compact_block(out.get(n))
    out['summary'] = b.summary_of(out, blocks)
    line = json.dumps(out)
    assert len(line) < 7000, len(line)
    assert list(out)[-1] == 'summary' and line.rstrip('}').count('"summary"') == 1
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in out['roofline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in out['cpu_baseline']
    for n in ('cfg2_f32', 'cfg3_bf16', 'cfg2_bf16', 'cfg5_f32', 'resblock1_f32', 'resblock1_bf16', 'train_step', 'stat_sync'):
        assert n in out['summary'], n
    assert len(line[-2048:].split('"summary"')) == 2       # the summary sits inside the last 2 KB


def test_launch_plan_tape_patches_bound_pointers_and_keeps_order():
    """schedule.Tape (the recorded launch plan of a forward, host logic only): a recorded call's pointer slots that held the address of a bound
    tensor - a plain integer argument, a c_void_p field of a struct passed by reference, an element of a pointer array inside it - are rewritten
    at every replay, everything else is left alone; calls go out in recorded order on the stream they were recorded on; a bound input nobody
    reads is an error (the plan would silently ignore it)."""
    import ctypes as C
    from wavthruvec_pytorch_amd import schedule

    class Args(C.Structure):
        _fields_ = [('in_', C.c_void_p), ('w', C.c_void_p), ('outs', C.c_void_p * 3), ('n', C.c_int32)]

    class FakeStream:
        def __init__(self, h):
            self.cuda_stream = h

    calls = []

    def fake(name):
        def fn(*a):
            calls.append((name,) + tuple(getattr(x, '_obj', x) for x in a))
            return 0
        return fn

    X, W, Y, NZ = 0x1000, 0x2000, 0x3000, 0x4000
    a1 = Args(); a1.in_, a1.w, a1.n = X, W, 7
    a1.outs[0], a1.outs[1], a1.outs[2] = 0x5000, Y, 0x6000
    a2 = Args(); a2.in_, a2.w, a2.n = NZ, W, 9
    tape = schedule.Tape()
    tape.steps.append(schedule.Step(schedule.K_CALL, fake('first'), [C.byref(a1)], schedule.MAIN, 'conv_pre', 'first'))
    tape.steps.append(schedule.Step(schedule.K_CALL, fake('second'), [C.byref(a2), X], schedule.SIDE, None, 'second'))
    tape.finalize({'x': X, 'nz': NZ, 'y': Y})
    assert tape.launches == 2 and len(tape.patches['x']) == 2 and len(tape.patches['nz']) == 1 and len(tape.patches['y']) == 1
    main, side = FakeStream(11), FakeStream(22)
    tape.replay(main, side, {'x': 0xA000, 'nz': 0xB000, 'y': 0xC000})
    assert [c[0] for c in calls] == ['first', 'second'] and calls[0][-1] == 11 and calls[1][-1] == 22
    assert (a1.in_, a1.w, a1.n, a1.outs[0], a1.outs[1], a1.outs[2]) == (0xA000, W, 7, 0x5000, 0xC000, 0x6000)
    assert (a2.in_, a2.w, a2.n) == (0xB000, W, 9) and calls[1][2] == 0xA000
    tape.replay(main, side, {'x': 0xD000, 'nz': 0xE000, 'y': 0xF000})          # a second replay rebinds the same slots
    assert (a1.in_, a1.outs[1], a2.in_) == (0xD000, 0xF000, 0xE000) and calls[3][2] == 0xD000
    bad = schedule.Tape()
    bad.steps.append(schedule.Step(schedule.K_CALL, fake('only'), [C.byref(a2)], schedule.MAIN, None, 'only'))
    with pytest.raises(RuntimeError, match='no recorded call reads x'):
        bad.finalize({'x': 0x7777, 'y': Y})
    with pytest.raises(RuntimeError, match='share an address'):
        schedule.Tape().finalize({'x': 0x1, 'nz': 0x1})


def test_round5_host_only_queries_answer_without_a_gpu():
    """The shape queries added in round 5 are host-only (no stream, nothing dereferenced): the bf16 weight gradient's slab count and the row count
    of the backward stage kernel's bias-gradient partials - which shapes they take and which they decline."""
    lib = _hip.load()
    # v2w_wgrad_bf16_slabs(B, c_in, c_out, Lq, k): C_in == C_out in {16, 32} or a multiple of 64, k odd <= 11, Lq % 8 == 0
    for C_, L in ((16, 81920), (32, 40960), (64, 20480), (128, 5120), (256, 1280)):
        for k in (3, 7, 11):
            assert lib.v2w_wgrad_bf16_slabs(32, C_, C_, L, k) > 0, (C_, k)
    assert lib.v2w_wgrad_bf16_slabs(32, 48, 48, 1024, 3) == 0          # 48 channels
    assert lib.v2w_wgrad_bf16_slabs(32, 64, 32, 1024, 3) == 0          # C_in != C_out
    assert lib.v2w_wgrad_bf16_slabs(32, 32, 32, 1028, 3) == 0          # rows not 16-byte aligned in bf16
    assert lib.v2w_wgrad_bf16_slabs(32, 32, 32, 1024, 4) == 0 and lib.v2w_wgrad_bf16_slabs(32, 32, 32, 1024, 13) == 0
    assert lib.v2w_wgrad_bf16_slabs(2, 16, 16, 64, 3) <= 2             # never more splits than staged items
    # v2w_resblock2_stage_bwd_rows: one row per (tile, wave) of the launcher's own geometry (conv2's dilations first)
    a = _hip.StageArgs()
    a.nk, a.B, a.C, a.L = 3, 32, 32, 40960
    for j, k in enumerate((3, 7, 11)):
        a.k[j], a.dil1[j], a.dil2[j] = k, 3, 1
    nto = (256 - 2 * 5) & ~3
    assert lib.v2w_resblock2_stage_bwd_rows(ctypes.byref(a)) == 32 * ((40960 + nto - 1) // nto) * 4
    a.C = 64
    assert lib.v2w_resblock2_stage_bwd_rows(ctypes.byref(a)) == 0      # wide stages have no one-kernel backward


def test_generator_copies_and_pickles_without_its_launch_plans():
    """Generator._tapes holds ctypes function pointers, byref objects and pointer-bearing structs (schedule.Tape); workspaces and
    fold caches hold device tensors and descriptor tables of THIS module.  copy.deepcopy / pickle (the EMA or snapshot pattern) must not
    raise on them and must not hand them to the copy - which would replay into the original's buffers."""
    import copy
    import ctypes
    import pickle
    from wavthruvec_pytorch_amd import Generator, synthetic

    h = synthetic.make_hparams(num_wv_feat=768)
    g = Generator(h)
    g.load_state_dict(synthetic.make_state_dict(h, seed=0))

    class Args(ctypes.Structure):
        _fields_ = [('p', ctypes.c_void_p), ('n', ctypes.c_int)]

    arg = Args(12345, 3)
    fake_tape = dict(fn=ctypes.CFUNCTYPE(ctypes.c_int)(lambda: 0), args=(ctypes.byref(arg), arg, ctypes.pointer(arg)))
    with pytest.raises((ValueError, TypeError)):
        copy.deepcopy(fake_tape)                                 # what a tape is made of does not copy
    g._tapes[('key',)] = fake_tape
    g._ws['act.pre'] = torch.zeros(4)
    g._fold_key.update(state=('x',), plan=fake_tape, gen=3)
    g._ws_epoch = 7
    g.precision = 'bf16'
    for clone in (copy.deepcopy(g), pickle.loads(pickle.dumps(g))):
        assert clone._tapes == {} and clone._ws == {} and clone._fold_key == {} and clone._slabs == {} and clone._ws_epoch == 0
        assert clone.precision == 'bf16' and clone.use_launch_plan
        for (ka, va), (kb, vb) in zip(g.state_dict().items(), clone.state_dict().items()):
            assert ka == kb and torch.equal(va, vb)
        assert clone.conv_pre.weight_v.data_ptr() != g.conv_pre.weight_v.data_ptr()
    assert g._tapes and g._ws and g._ws_epoch == 7               # the original keeps its own


def test_kernel_name_sink_reports_what_a_call_would_launch():
    """ABI v33: a launching entry point handed a v2w_name_sink in place of the stream runs its checks and kernel selection, launches nothing and
    appends the demangled kernel names.  Host-only (no GPU here): the residual stages on bf16 tensors, with and without the fused tail / the
    fused upsampler, and a conv.  bench.py labels its rooflines with these names - it holds no kernel-name table of its own."""
    import ctypes as C
    import re
    lib = _hip.load()

    def stage(Cc, L, post=False, up=0):
        a = _hip.StageSplitArgs()
        for j, k in enumerate((3, 7, 11)):
            a.k[j], a.dil1[j], a.dil2[j] = k, 1, 3
            a.wps1[j], a.wps2[j], a.sc1[j], a.sc2[j] = 0x10000, 0x20000, 0x30000, 0x30000      # (aligned, never dereferenced)
        a.in_, a.nk, a.B, a.C, a.L = 0x100000, 3, 2, Cc, L
        a.slope, a.out_div, a.bf16, a.io_bf16 = 0.1, 3.0, 1, 3
        if post:
            a.post_w, a.post_out, a.post_k, a.post_slope = 0x5000, 0x600000, 7, 0.01
        elif up:
            a.up_wps, a.up_out, a.up_k, a.up_u, a.up_slope = 0x8000, 0x900000, 2 * up, up, 0.1
        else:
            a.out = 0x700000
        return a

    got = {}
    for key, a in dict(c16_tail=stage(16, 4096, post=True), c16=stage(16, 4096), c32=stage(32, 4096), c128_up4=stage(128, 4096, up=4),
                       c256=stage(256, 1024)).items():
        rc, names = _hip.kernel_names(lib.v2w_resblock2_stage_split_fwd, C.byref(a))
        assert rc in (0, 100), (key, rc)            # (100 = hipErrorNoDevice from hipGetLastError in a GPU-less process: nothing was launched)
        assert len(names) == 1, (key, names)
        got[key] = names[0]
    assert got['c16_tail'] == 'n16s_stage_kernel' and got['c16'].startswith('n16_stage_kernel<')
    assert all(got[k].startswith('wide_stage_bf16_kernel<') for k in ('c32', 'c128_up4', 'c256')) and len(set(got.values())) == 5
    c = _hip.Conv1dArgs()
    c.in_, c.out, c.wp = 0x1000, 0x2000, 0x3000
    c.B, c.C_in, c.C_out, c.L, c.k, c.dil, c.slope = 32, 768, 512, 256, 7, 1, 1.0
    rc, names = _hip.kernel_names(lib.v2w_conv1d_fwd, C.byref(c), short=False)
    assert rc in (0, 100) and len(names) == 1
    assert re.fullmatch(r'void \(anonymous namespace\)::conv_tile_kernel<[^()]*>\(\(anonymous namespace\)::MultiArgs\)', names[0]), names
    assert _hip.kernel_name_short(names[0]).startswith('conv_tile_kernel<32, 1,')
    # a sink too small for a name drops it and stays terminated
    buf = C.create_string_buffer(16)
    sink = _hip.NameSink(_hip.NAME_SINK_MAGIC, C.addressof(buf), len(buf), 0)
    lib.v2w_conv1d_fwd(C.byref(c), C.c_void_p(C.addressof(sink) | 1))
    assert sink.len == 0 and buf.value == b''
    # bench.py asks the library: no kernel-name literal, no template-argument table
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert not re.search(r"_kernel\s*<|[a-z0-9]_kernel\b(?!s)", src.replace('per_kernel', '').replace('sum_conv_kernel_ms', '').replace('profile_kernel_names', ''))


def test_integration_md_stub_runs_against_the_library():
    """the ctypes stub INTEGRATION.md documents must work as written - it is executed here (up to the first device tensor), its
    ABI check against the header's constant, its struct against the binding the package uses."""
    import ctypes as C
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = text[text.index('```python', text.index('## 2. C-ABI level')) + len('```python'):]
    block = block[:block.index('```')]
    host_part = block[:block.index('# x + conv')]
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        exec(compile(host_part, 'INTEGRATION.md', 'exec'), ns)
    finally:
        os.chdir(cwd)
    assert ns['abi'] == _hip.ABI_VERSION
    stub, mirror = ns['Conv1dArgs'], _hip.Conv1dArgs
    assert C.sizeof(stub) == C.sizeof(mirror)
    assert [(n, getattr(stub, n).offset, getattr(stub, n).size) for n, _t in stub._fields_] == \
           [(n, getattr(mirror, n).offset, getattr(mirror, n).size) for n, _t in mirror._fields_]


def test_tape_refuses_pointers_the_module_does_not_own():
    """a launch plan replays prebuilt structs; a struct slot that points into memory nobody keeps alive (a scratch tensor a wrapper
    allocated inside the planned forward) must keep the plan from being stored.  Host-only: ranges and slots are plain integers."""
    import ctypes as C
    from wavthruvec_pytorch_amd import schedule

    assert schedule._in_ranges([(100, 200), (300, 400)], 150) and schedule._in_ranges([(100, 200), (300, 400)], 300)
    assert not schedule._in_ranges([(100, 200), (300, 400)], 200) and not schedule._in_ranges([(100, 200)], 50) and not schedule._in_ranges([], 1)

    class A(C.Structure):
        _fields_ = [('in_', C.c_void_p), ('w', C.c_void_p * 2), ('out', C.c_void_p), ('n', C.c_int32)]

    def tape_with(a):
        t = schedule.Tape()
        t.steps.append(schedule.Step(schedule.K_CALL, fn=None, args=[C.byref(a)], name='v2w_fake'))
        return t

    owned = [(0x1000, 0x2000), (0x8000, 0x9000)]
    binds = dict(x=0x5000, y=0x6000)
    ok = A(0x5000, (C.c_void_p * 2)(0x1100, 0x8800), 0x6000, 7)
    t = tape_with(ok)
    t.finalize(binds, owned=owned)
    assert len(t.patches['x']) == 1 and len(t.patches['y']) == 1 and t.launches == 1
    bad = A(0x5000, (C.c_void_p * 2)(0x1100, 0x7000), 0x6000, 7)       # 0x7000: nobody's
    with pytest.raises(schedule.TapeNotOwned):
        tape_with(bad).finalize(binds, owned=owned)
    tape_with(bad).finalize(binds)                                        # (no ranges given: the round-5 behaviour)
    null = A(0x5000, (C.c_void_p * 2)(0x1100, 0), 0x6000, 7)              # NULL slots are optional arguments
    tape_with(null).finalize(binds, owned=owned)


def test_bench_schedule_roofline_uses_the_cited_profile_or_says_why_not(tmp_path, monkeypatch):
    """bench.py `schedule`: counter bytes of the step's kernels (profiles/rNN_<suffix>, FETCH x 2 + WRITE per launch) x launches / step time /
    8 TB/s - and an explicit `missing` entry, never a silent skip, when a kernel the library names has no row in the profile it cites or the
    profile was collected from other kernel sources."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    prof = tmp_path / 'profiles'
    prof.mkdir()
    sha = bench.csrc_hash()
    (prof / 'r99_fake_hbm_traffic.json').write_text(json.dumps({
        '_meta': dict(commit='abc', date='today', csrc_sha=sha),
        'kernel_a<1, 2>': dict(launches=3, hbm_bytes_per_launch=4e8),
        'kernel_b': dict(launches=3, hbm_bytes_per_launch=1e8)}))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    per_kernel = {'kernel_a<1, 2>': dict(launches=2, ms=1.0, tflops=1.0), 'kernel_b': dict(launches=1, ms=0.5, tflops=1.0)}
    sc = bench.schedule_traffic(per_kernel, 'fake_hbm_traffic.json', 1e-3)
    assert sc['counter_bytes_per_step'] == 9e8 and abs(sc['hbm_frac_counter'] - 9e8 / 1e-3 / 1e9 / bench.PEAK_HBM_GBS) < 1e-12
    sc = bench.schedule_traffic(dict(per_kernel, **{'kernel_c + kernel_b': dict(launches=1, ms=0.1, tflops=0.0)}), 'fake_hbm_traffic.json', 1e-3)
    assert sc['hbm_frac_counter'] is None and len(sc['missing']) == 1 and 'kernel_c' in sc['missing'][0]
    c = bench._compact_schedule(sc)
    assert c['missing'][0] == 1 and len(json.dumps(c)) < 200
    (prof / 'r99_fake_hbm_traffic.json').write_text(json.dumps({'_meta': dict(csrc_sha='other'), 'kernel_a<1, 2>': dict(launches=1, hbm_bytes_per_launch=1.0),
                                                                'kernel_b': dict(launches=1, hbm_bytes_per_launch=1.0)}))
    sc = bench.schedule_traffic(per_kernel, 'fake_hbm_traffic.json', 1e-3)
    assert sc['hbm_frac_counter'] is None and 'stale' in sc['missing'][0]
    assert bench.schedule_traffic(per_kernel, None, 1e-3) is None


def test_planner_waits_once_per_side_stream_and_never_for_more_than_asked():
    """forward_plan.ForwardPlanner.need (round 6): the event behind a step of an in-order stream stands for the earlier steps of THAT stream -
    the latest of the asked steps is waited for, the earlier ones count as met and are never waited for again; steps of another stream keep
    their own wait; a later step is never waited for on behalf of an earlier one."""
    import types
    import torch
    from wavthruvec_pytorch_amd.forward_plan import ForwardPlanner

    class FakeStreams:
        def __init__(self):
            self.log = []

        def mark(self, name, sid=1):
            self.log.append(('mark', name, sid))

        def need(self, name):
            self.log.append(('need', name))

    for merge in (True, False):
        g = types.SimpleNamespace(training=True, algo=0, num_kernels=3, num_upsamples=5, merge_waits=merge)
        S = FakeStreams()
        pl = ForwardPlanner(g, torch.zeros(2, 3, 4), None, None, None, S)
        for name, sid in (('post', 1), ('ups.0', 1), ('cond', 2), ('rest', 1), ('ups.1', 1), ('ups.2', 1)):
            pl.mark(name, sid)
        pl.need('ups.0')
        pl.need('cond')
        pl.need('rest', 'ups.1')
        pl.need('post')                  # behind ups.0 on its stream
        pl.need('ups.2', 'nothing-marked-under-this-name')
        pl.need('ups.2')                 # each step once
        waits = [e[1] for e in S.log if e[0] == 'need']
        if merge:
            assert waits == ['ups.0', 'cond', 'ups.1', 'ups.2']
        else:
            assert waits == ['ups.0', 'cond', 'rest', 'ups.1', 'post', 'ups.2']
        assert pl.needed >= {'post', 'ups.0', 'cond', 'rest', 'ups.1', 'ups.2'}
