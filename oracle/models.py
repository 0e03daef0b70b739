"""torch-CPU restatement of the reference's three model builders (models.py:8-89).

TEST INFRASTRUCTURE (see oracle/__init__.py) - parity unpinned.

Parameters live in a plain ``dict`` keyed by the slim variable names of SURVEY Appendix C
(``g/conv1/weights``, ``g/conv1/BatchNorm/beta``, ``g/tconv4/biases`` ...).  The layer
tables below restate the topology; the slim layer contract (conv -> BN | +bias ->
activation) is applied by ``_layer``.

Documented deviations from the reference text (it does not run as committed):
  D2  slim.argscope -> slim.arg_scope defaults cascade as written.
  D3  D's action tile matches the conv2 output size (H/4), not 4x4.
  D9  ksize is a parameter (default 5); H, W are taken from the tensor.
  128x128: action tile is H/16, sconv5 kernel is H/16 (build decision, no reference behaviour).
"""
import torch

from . import tf_ops as T

# (scope, kind, cout, ksize, stride, padding, norm, act)
#   kind: 'c' conv2d, 't' conv2d_transpose;  norm: True = slim.batch_norm, False = bias


def generator_layers():
    """models.py:12-21."""
    return {
        'enc': [('conv1', 'c', 64, 5, 2, 'SAME', True, 'relu'),
                ('conv2', 'c', 128, 5, 2, 'SAME', True, 'relu'),
                ('conv3', 'c', 256, 5, 2, 'SAME', True, 'relu'),
                ('conv4', 'c', 512, 5, 2, 'SAME', True, 'relu')],
        'dec': [('tconv1', 't', 256, 5, 2, 'SAME', True, 'relu'),
                ('tconv2', 't', 128, 5, 2, 'SAME', True, 'relu'),
                ('tconv3', 't', 64, 5, 2, 'SAME', True, 'relu'),
                ('tconv4', 't', 3, 5, 2, 'SAME', False, 'tanh')],
    }


def generator_transform_layers(ksize, state_k):
    """models.py:34-59."""
    return {
        'enc': [('conv1', 'c', 32, 5, 2, 'SAME', True, 'relu'),
                ('conv2', 'c', 64, 5, 2, 'SAME', True, 'relu'),
                ('conv3', 'c', 128, 5, 2, 'SAME', True, 'relu'),
                ('conv4', 'c', 256, 5, 2, 'SAME', True, 'relu')],
        'dec1': [('tconv1', 't', 128, 5, 2, 'SAME', True, 'relu'),
                 ('tconv2', 't', 128, 5, 2, 'SAME', True, 'relu')],
        'state': [('sconv3', 'c', 32, 3, 2, 'SAME', True, 'relu'),
                  ('sconv4', 'c', 16, 3, 2, 'SAME', True, 'relu'),
                  ('sconv5', 'c', 5, state_k, 1, 'VALID', False, None)],
        'dec2': [('tconv3', 't', 128, 5, 2, 'SAME', True, 'relu'),
                 ('tconv4', 't', ksize * ksize, 5, 2, 'SAME', False, None)],
    }


def discriminator_layers():
    """models.py:82-88 (conv6 keeps the argscope BN, no activation)."""
    return {
        'pre': [('conv1', 'c', 64, 5, 2, 'SAME', True, 'lrelu'),
                ('conv2', 'c', 128, 5, 2, 'SAME', True, 'lrelu')],
        'post': [('conv3', 'c', 128, 5, 2, 'SAME', True, 'lrelu'),
                 ('conv4', 'c', 256, 5, 2, 'SAME', True, 'lrelu'),
                 ('conv5', 'c', 512, 5, 2, 'SAME', True, 'lrelu'),
                 ('conv6', 'c', 1, 2, 1, 'SAME', True, None)],
    }


_ACT = {'relu': T.relu, 'lrelu': T.lrelu, 'tanh': torch.tanh, None: lambda x: x}


def _layer(params, net, spec, x, create=None):
    scope, kind, cout, k, s, pad, norm, act = spec
    cin = x.shape[-1]
    wname = '%s/%s/weights' % (net, scope)
    if create is not None and wname not in params:
        gen, dtype = create
        shape = (k, k, cin, cout) if kind == 'c' else (k, k, cout, cin)
        # slim xavier: fan_in = kh*kw*shape[-2], fan_out = kh*kw*shape[-1]
        params[wname] = T.xavier_uniform_(shape, k * k * shape[2], k * k * shape[3], gen, dtype)
        bname = '%s/%s/%s' % (net, scope, 'BatchNorm/beta' if norm else 'biases')
        params[bname] = torch.zeros(cout, dtype=dtype)
    # T.q_act / T.q_weight are identities unless bf16 storage is being emulated (tf_ops.bf16_storage): then the conv
    # reads bf16 operands and writes a bf16 tensor, a BatchNorm'd + activated layer output is bf16 again, and the heads
    # (bias layers, d/conv6 without activation) hand float32 to the losses
    w = T.q_weight(params[wname])
    x = T.q_act(x)
    y = T.conv2d(x, w, s, pad) if kind == 'c' else T.conv2d_transpose(x, w, s, pad)
    # a BatchNorm'd layer without activation (d/conv6) keeps its conv output in float32, only its gradient is bf16
    y = T.q_grad(y) if (norm and act is None) else T.q_act(y)
    if norm:
        y = T.batch_norm_train(y, params['%s/%s/BatchNorm/beta' % (net, scope)])
    else:
        y = y + params['%s/%s/biases' % (net, scope)]
    out = _ACT[act](y)
    return T.q_act(out) if (norm and act is not None) else out


def tile_actions(actions, size):
    """train.py:48-50: [B,A] -> [B,size,size,A]."""
    b, a = actions.shape
    return actions.reshape(b, 1, 1, a).expand(b, size, size, a)


def generator(params, images, actions, create=None):
    """models.py:8-22.  images [B,H,W,3], actions [B,A] (tiled here to H/16)."""
    L = generator_layers()
    out = images
    for spec in L['enc']:
        out = _layer(params, 'g', spec, out, create)
    out = T.q_act(torch.cat([out, tile_actions(actions, out.shape[1]).to(out.dtype)], dim=3))
    for spec in L['dec']:
        out = _layer(params, 'g', spec, out, create)
    return out


def generator_transform(params, images, actions, ksize=5, create=None):
    """models.py:24-74.  Returns (frame [B,H,W,3], state [B,5])."""
    L = generator_transform_layers(ksize, images.shape[1] // 16)
    out = images
    for spec in L['enc']:
        out = _layer(params, 'g', spec, out, create)
    out = T.q_act(torch.cat([out, tile_actions(actions, out.shape[1]).to(out.dtype)], dim=3))
    for spec in L['dec1']:
        out = _layer(params, 'g', spec, out, create)
    st = out
    for spec in L['state']:
        st = _layer(params, 'g', spec, st, create)
    for spec in L['dec2']:
        out = _layer(params, 'g', spec, out, create)
    frame = T.dna_gather(out, images, ksize)
    return frame, st.reshape(st.shape[0], -1)


def discriminator(params, inputs, actions, create=None):
    """models.py:76-89 with D3 resolved.  inputs [B,H,W,6], actions [B,A] -> logits [B,h,w,1]."""
    L = discriminator_layers()
    out = inputs
    for spec in L['pre']:
        out = _layer(params, 'd', spec, out, create)
    out = T.q_act(torch.cat([out, tile_actions(actions, out.shape[1]).to(out.dtype)], dim=3))
    for spec in L['post']:
        out = _layer(params, 'd', spec, out, create)
    return out


def init_params(arg_transform, batch=2, img=64, ksize=5, seed=0, dtype=torch.float32, act_dim=10):
    """Create all g/ and d/ variables (slim xavier-uniform weights, zero beta/bias)."""
    gen = torch.Generator().manual_seed(seed)
    params = {}
    x = torch.zeros(batch, img, img, 3, dtype=dtype)
    a = torch.zeros(batch, act_dim, dtype=dtype)
    with torch.no_grad():
        if arg_transform:
            frame, _ = generator_transform(params, x, a, ksize, create=(gen, dtype))
        else:
            frame = generator(params, x, a, create=(gen, dtype))
        discriminator(params, torch.cat([x, frame], dim=3), a, create=(gen, dtype))
    return params
