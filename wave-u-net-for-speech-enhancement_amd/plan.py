"""Channel plan of the Wave-U-Net (mirror of the reference Model.__init__,
/root/reference/model/unet_basic.py:33-75) used by the module and the engine."""


def conv_layer_shapes(n_layers=12, channels_interval=24):
    """[(c_in, c_out, taps)] for encoder.0..n-1, middle, decoder.0..n-1 (forward order)."""
    n, ci = n_layers, channels_interval
    shapes = [((1 if i == 0 else i * ci), (i + 1) * ci, 15) for i in range(n)]
    shapes.append((n * ci, n * ci, 15))
    for j in range(n):
        c_in = 2 * n * ci if j == 0 else (2 * (n - j) + 1) * ci
        shapes.append((c_in, (n - j) * ci, 5))
    return shapes
