"""Generator / discriminator builders with the reference's signatures (models.py:8, 24-29, 76-78).

Each network is a short table of (scope, width) rows walked under an ``arg_scope`` that carries the
slim defaults the reference sets (stride 2, SAME, batch_norm, relu / lrelu); the rows that deviate
(heads without BatchNorm, the VALID state head, d/conv6 without activation) override per call.
Variables are named exactly as slim names them (``g/conv1/weights``, ``g/conv1/BatchNorm/beta``,
``g/tconv4/biases``), so ``reuse=True`` shares them across calls.

Deviations from the reference text, which does not run as committed:
  D2  ``slim.argscope`` is read as ``slim.arg_scope``.
  D3  the discriminator tiles its actions to its own conv2 output size, not to 4x4.
  D9  ``ksize`` is honoured (default 5) and H, W come from the tensor instead of a literal 64.
``images.padded`` (optional attribute): the same frames stored with a channel pitch of 4; g/conv1 then gathers
with 16-byte loads.  Actions may be given as ``[B, A]`` (tiled and concatenated in one fused op) or already tiled
``[B, h, w, A]`` as the reference Trainer passes them (train.py:48-50).
"""
from . import graph as G
from . import ops as O

G_PLAIN = {'encoder': (('conv1', 64), ('conv2', 128), ('conv3', 256), ('conv4', 512)),
           'decoder': (('tconv1', 256), ('tconv2', 128), ('tconv3', 64))}
G_DNA = {'encoder': (('conv1', 32), ('conv2', 64), ('conv3', 128), ('conv4', 256)),
         'decoder_a': (('tconv1', 128), ('tconv2', 128)),
         'state': (('sconv3', 32), ('sconv4', 16)),
         'decoder_b': (('tconv3', 128),)}
D_NET = {'pre': (('conv1', 64), ('conv2', 128)), 'post': (('conv3', 128), ('conv4', 256), ('conv5', 512))}


def _with_actions(features, actions, name):
    """Channel-concatenate the action/state vector, broadcast over the feature map."""
    if len(actions.shape) == 2:
        return O.concat_actions(features, actions, name=name)
    if len(actions.shape) == 4 and actions.shape[1:3] == features.shape[1:3]:
        if O.half_mode():
            raise NotImplementedError('bf16 graphs take the actions as [B, A] (tiled and concatenated in one fused op)')
        return O.concat([features, actions], axis=3, name=name)
    raise ValueError('actions %s cannot be concatenated with a %s feature map' % (actions.shape, features.shape))


def _stack(net, rows, layer, size=5):
    for scope, width in rows:
        net = layer(net, width, [size, size], scope=scope)
    return net


def build_generator(images, actions, reuse=False):
    """Plain generator (models.py:8-22): 4 conv down, actions, 3 deconv up, deconv + bias + tanh."""
    with O.variable_scope('g', reuse=reuse), \
            O.arg_scope([O.conv2d, O.deconv2d], activation_fn=O.relu, stride=2, padding='SAME',
                        normalizer_fn=O.batch_norm, reuse=reuse):
        net = _stack(getattr(images, 'padded', images), G_PLAIN['encoder'], O.conv2d)
        net = _with_actions(net, actions, 'actions')
        net = _stack(net, G_PLAIN['decoder'], O.deconv2d)
        return O.deconv2d(net, images.shape[3], [5, 5], activation_fn=O.tanh, normalizer_fn=None, scope='tconv4')


def build_generator_transform(images, actions, batch_size=None, reuse=False, color_channels=3, ksize=5):
    """DNA generator (models.py:24-74): predicts k*k per-pixel kernel logits and a 5-d next state.

    Returns ``(frame, state)``; ``batch_size`` is accepted for signature compatibility (the static
    shape already carries it).
    """
    if batch_size is not None and batch_size != images.shape[0]:
        raise ValueError('batch_size %r does not match images %s' % (batch_size, images.shape))
    if images.shape[3] != color_channels:
        raise ValueError('color_channels %r does not match images %s' % (color_channels, images.shape))
    with O.variable_scope('g', reuse=reuse), \
            O.arg_scope([O.conv2d, O.deconv2d], activation_fn=O.relu, stride=2, padding='SAME',
                        normalizer_fn=O.batch_norm, reuse=reuse):
        net = _stack(getattr(images, 'padded', images), G_DNA['encoder'], O.conv2d)
        net = _with_actions(net, actions, 'actions')
        net = _stack(net, G_DNA['decoder_a'], O.deconv2d)

        # the state head reads the decoder's 16x16 features and nothing of what the frame decoder does next: a side chain
        # (five tiny latency-bound layers that hide behind tconv3 / tconv4 / the DNA gather, forward and backward)
        with G.get_default_graph().side_branch():
            state = _stack(net, G_DNA['state'], O.conv2d, size=3)
            sk = state.shape[1]           # 4 at 64x64: the reference's 4x4 VALID head (models.py:44-51)
            state = O.conv2d(state, 5, [sk, sk], activation_fn=None, stride=1, padding='VALID', normalizer_fn=None,
                             scope='sconv5')
            state = O.squeeze(state)

        net = _stack(net, G_DNA['decoder_b'], O.deconv2d)
        logits = O.deconv2d(net, ksize * ksize, [5, 5], activation_fn=None, normalizer_fn=None, scope='tconv4')
        frame = O.dna_gather(logits, images, ksize)
        return frame, state


def build_discriminator(inputs, actions, reuse=False):
    """Discriminator (models.py:76-89): 5 x [conv5x5/2 + BN + lrelu], actions after conv2, 2x2 logit conv + BN."""
    with O.variable_scope('d', reuse=reuse), \
            O.arg_scope([O.conv2d], activation_fn=O.lrelu, stride=2, padding='SAME', normalizer_fn=O.batch_norm,
                        reuse=reuse):
        net = _stack(inputs, D_NET['pre'], O.conv2d)
        if len(actions.shape) == 4 and actions.shape[1:3] != net.shape[1:3]:
            raise ValueError('discriminator: actions tiled to %s but conv2 output is %s (reference defect D3: '
                             'pass [B, A] actions or tile to H/4)' % (actions.shape[1:3], net.shape[1:3]))
        net = _with_actions(net, actions, 'actions')
        net = _stack(net, D_NET['post'], O.conv2d)
        return O.conv2d(net, 1, [2, 2], activation_fn=None, stride=1, scope='conv6')
