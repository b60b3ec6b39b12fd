"""The C ABI's resource promises (include/hoig_kernels.h: the caller owns every buffer, nothing behind the interface allocates) and the
host's side of them: a process that builds Trainer after Trainer keeps a flat number of HIP streams and
a flat amount of device memory; the thin-channel weight gradients reduce through a scratch block the CALLER registered and say so
when there is none."""
import ctypes
import gc

import pytest
import torch

from common import product_trainer

pytestmark = pytest.mark.gpu


def _census():
    from hoig_amd import ops
    gc.collect()
    torch.cuda.synchronize()
    return ops.stream_census()


def test_trainers_come_and_go_with_a_flat_stream_and_memory_count():
    from hoig_amd import ops
    ops.set_precision('bf16x3:f16x2')
    try:
        def one():
            m = product_trainer('generator_spade_attn', 1, 64)
            m.optimize_parameters()
            m.optimize_parameters()
            err = m.get_current_errors()
            assert all(v == v for v in err.values())
            m.close()
            del m
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        one()
        live0, idle0, scratch0 = _census()
        mem0 = torch.cuda.memory_allocated()
        assert scratch0 > 0, 'a training step launches thin-channel weight gradients: their stream registers a scratch block'
        for _ in range(4):
            one()
        live1, idle1, scratch1 = _census()
        assert (live1, idle1) == (live0, idle0), 'streams made and not handed back: %r -> %r' % ((live0, idle0), (live1, idle1))
        assert scratch1 == scratch0
        assert torch.cuda.memory_allocated() <= mem0 + (1 << 20), (mem0, torch.cuda.memory_allocated())
        # everything that sits in the banks can be destroyed; what stays is what is still in use (the module's weight-gradient stream,
        # streams of objects earlier tests of the session left alive)
        ops.destroy_idle_streams()
        live2, idle2, _ = _census()
        assert idle2 == 0 and live2 == live1 - idle1
        one()                                            # and the banks refill
        assert _census()[0] >= live2
    finally:
        ops.set_precision('f32')


def test_thin_weight_gradient_uses_the_registered_scratch_and_survives_without():
    """hoig_conv2d_bwd_weight on a thin-input layer (7x7, 8 -> 64: the generator's stem): with the stream's scratch block registered the
    partials are reduced through it; with none the kernel takes its atomic path -- same sums up to the order of additions."""
    from hoig_amd import _lib as L
    from hoig_amd import ops
    B, S, Ci, Co = 2, 64, 8, 64
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(B, S, S, Ci, device='cuda', generator=g)
    dy = torch.randn(B, S, S, Co, device='cuda', generator=g)
    d = L.ConvDesc(B, S, S, Ci, S, S, Co, 7, 7, 1, 3, 0, L.ACT_NONE, 0.0, L.PREC_F16X2)
    s = torch.cuda.Stream()
    outs = []
    with torch.cuda.stream(s):
        for registered in (True, False):
            dw = torch.zeros(Co, 7, 7, Ci, device='cuda')
            if registered:
                ops.wgrad_call('hoig_conv2d_bwd_weight', d, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, s.cuda_stream)
            else:
                L.call('hoig_stream_scratch_set', s.cuda_stream, None, 0)
                L.call('hoig_conv2d_bwd_weight', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, s.cuda_stream)
            outs.append(dw)
        ops._scratch.pop((x.device, s.cuda_stream), None)
    s.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.cpu().double().permute(0, 3, 1, 2), (Co, Ci, 7, 7), dy.cpu().double().permute(0, 3, 1, 2),
                                      padding=3).permute(0, 2, 3, 1).float().cuda()
    for dw in outs:
        rel = ((dw - ref).norm() / ref.norm()).item()
        assert rel < 3e-3, rel                            # two-term bf16 arithmetic
    assert ((outs[0] - outs[1]).norm() / ref.norm()).item() < 1e-5


def test_a_bare_network_reads_a_checkpoint_saved_through_ddp_in_both_dataset_copies(tmp_path):
    """`need_module=True` is how the HOIG_DexYCB copy feeds its DDP wrappers (its trainer.py:562,566); a single-GPU run of
    that copy must still load the 'module.'-prefixed files a DDP run saved."""
    from hoig_amd.models import ModelsFactory
    from common import opt_namespace
    for dataset in ('dexycb', 'hov3'):
        d = tmp_path / dataset
        m = product_trainer('generator_spade', 1, 64, dataset=dataset, checkpoints_dir=str(d))
        m.save(3)
        for net in ('G', 'D'):
            path = m._ckpt.file('net', 3, net)
            sd = torch.load(path, map_location='cpu')
            torch.save(type(sd)(('module.' + k, v) for k, v in sd.items()), path)
        m2 = ModelsFactory.get_by_name('trainer', opt_namespace(checkpoints_dir=str(d), load_epoch=3, dataset_mode=dataset,
                                                                 gen_name='generator_spade'))
        for k, v in m._G.state_dict().items():
            assert torch.equal(v, m2._G.state_dict()[k]), k
        m.close(); m2.close()
