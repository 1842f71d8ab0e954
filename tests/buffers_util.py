"""Buffer-by-buffer comparison of an engine with the NumPy interpreter, with and without the chained two-layer MLP
kernel (option "mlp_fuse"): chained, the hidden activations of such an MLP never reach HBM, so their buffers are
compared only in the unchained run; everything downstream is compared in both."""
import numpy as np


def check_every_buffer(eng, it, B, evaluate, rtol=1e-9, atol=1e-9):
    """evaluate() runs the Laplacian-mode evaluation (and returns whatever the caller wants from the LAST, chained run)."""
    worst, out = {}, None
    for fuse in (0, 1):
        eng.set_option('mlp_fuse', fuse)
        out = evaluate()
        for name, idx in eng.program.buf_names.items():
            if fuse and '/hidden_' in name:
                continue
            got = eng.debug_read(name, B)
            worst[name] = max(worst.get(name, 0.0), float(np.abs(got - it.bufs[idx]).max()))
            np.testing.assert_allclose(got, it.bufs[idx], rtol=rtol, atol=atol, err_msg=f'buffer {name} (mlp_fuse={fuse})')
    return out, worst
