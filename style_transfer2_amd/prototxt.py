"""Network definition files: the subset of Caffe's ``.prototxt`` text format the reference's model uses
(reference config.ini:28 ``prototxt = models/vgg19.prototxt``; worker.py:58-61 ``caffe.Net(prototxt, 1, weights=...)``;
models/vgg19.prototxt:1-337) turned into the engine's topology ``(('conv', name, cin, cout) | ('pool', name), ...)``.

What the engine runs is what that file describes and nothing else: a chain of 3x3 / pad 1 / stride 1 convolutions, each
followed by an in-place ReLU, and 2x2 / stride 2 MAX pools.  Any other layer, parameter value or wiring is an error here
(loudly, at worker start) rather than a silently different network.  ``write`` emits the same subset -- the tests build
miniature networks with it; the stock VGG19 is built in (engine.VGG19_TOPOLOGY) and needs no file.
"""

import re

_TOKEN = re.compile(r'\s*(?:(#[^\n]*)|([{}:])|"((?:[^"\\]|\\.)*)"|([^\s{}:"#]+))')


def _tokens(text):
    pos, out = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip():
                raise ValueError('prototxt: cannot tokenise near %r' % text[pos:pos + 30])
            break
        pos = m.end()
        if m.group(1) is not None:
            continue
        if m.group(2) is not None:
            out.append(('p', m.group(2)))
        elif m.group(3) is not None:
            out.append(('s', m.group(3)))
        else:
            out.append(('w', m.group(4)))
    return out


def _message(tok, pos, until_brace):
    """{field: [values]}; a value is a string / bare word or a nested message dict."""
    msg = {}
    while pos < len(tok):
        kind, val = tok[pos]
        if kind == 'p' and val == '}':
            if not until_brace:
                raise ValueError('prototxt: unbalanced }')
            return msg, pos + 1
        if kind != 'w':
            raise ValueError('prototxt: field name expected, got %r' % (val,))
        name, pos = val, pos + 1
        if pos < len(tok) and tok[pos] == ('p', ':'):
            pos += 1
        if pos >= len(tok):
            raise ValueError('prototxt: value of %s missing' % name)
        if tok[pos] == ('p', '{'):
            sub, pos = _message(tok, pos + 1, True)
            msg.setdefault(name, []).append(sub)
        else:
            msg.setdefault(name, []).append(tok[pos][1])
            pos += 1
    if until_brace:
        raise ValueError('prototxt: missing }')
    return msg, pos


def _one(msg, key, default=None):
    vals = msg.get(key)
    if not vals:
        return default
    return vals[-1]


def parse(text):
    """Topology of a network definition.  The first layer must be the 3-channel ``Input``; blobs are chained top to
    bottom; ReLUs must be in place on the convolution above them."""
    net, _ = _message(_tokens(text), 0, False)
    topo, cur, channels, relu_pending = [], None, None, None
    for layer in net.get('layer', []):
        kind, name = _one(layer, 'type'), _one(layer, 'name')
        bottoms, tops = layer.get('bottom', []), layer.get('top', [])
        if kind == 'Input':
            dims = [int(d) for d in _one(_one(layer, 'input_param', {}), 'shape', {}).get('dim', [])]
            if len(dims) != 4 or dims[1] != 3:
                raise ValueError('prototxt: the input must be N x 3 x H x W, got %s' % dims)
            cur, channels = tops[0], 3
            continue
        if cur is None:
            raise ValueError('prototxt: layer %s comes before the Input layer' % name)
        if bottoms != [cur]:
            raise ValueError('prototxt: layer %s reads %s, the chain is at %s (only a linear chain is supported)' % (name, bottoms, cur))
        if relu_pending is not None and kind != 'ReLU':
            raise ValueError('prototxt: convolution %s is not followed by its in-place ReLU' % relu_pending)
        if kind == 'Convolution':
            p = _one(layer, 'convolution_param', {})
            k, pad, stride = int(_one(p, 'kernel_size', 0)), int(_one(p, 'pad', 0)), int(_one(p, 'stride', 1))
            if (k, pad, stride) != (3, 1, 1) or int(_one(p, 'group', 1)) != 1 or int(_one(p, 'dilation', 1)) != 1:
                raise ValueError('prototxt: %s is not a 3x3 / pad 1 / stride 1 convolution' % name)
            if tops != [name]:
                raise ValueError('prototxt: convolution %s must write a blob of its own name' % name)
            cout = int(_one(p, 'num_output'))
            topo.append(('conv', name, channels, cout))
            cur, channels, relu_pending = tops[0], cout, name
        elif kind == 'ReLU':
            if relu_pending is None or tops != [cur]:
                raise ValueError('prototxt: ReLU %s must be in place on the convolution above it' % name)
            if float(_one(_one(layer, 'relu_param', {}), 'negative_slope', 0)) != 0:
                raise ValueError('prototxt: leaky ReLU %s is not supported' % name)
            relu_pending = None
        elif kind == 'Pooling':
            p = _one(layer, 'pooling_param', {})
            if (_one(p, 'pool', 'MAX'), int(_one(p, 'kernel_size', 0)), int(_one(p, 'stride', 1)), int(_one(p, 'pad', 0))) != ('MAX', 2, 2, 0):
                raise ValueError('prototxt: %s is not a 2x2 / stride 2 MAX pool' % name)
            if tops != [name]:
                raise ValueError('prototxt: pool %s must write a blob of its own name' % name)
            topo.append(('pool', name))
            cur = tops[0]
        else:
            raise ValueError('prototxt: layer type %s (%s) is not supported' % (kind, name))
    if relu_pending is not None:
        raise ValueError('prototxt: convolution %s is not followed by its in-place ReLU' % relu_pending)
    if not topo:
        raise ValueError('prototxt: no layers')
    return tuple(topo)


def read(path):
    with open(path) as f:
        return parse(f.read())


def write(topology, name='net'):
    """Text of a definition ``parse`` maps back to ``topology`` (ReLU layers are named relu<suffix of the conv>)."""
    out = ['name: "%s"' % name, 'force_backward: true',
           'layer {\n    name: "data"\n    type: "Input"\n    top: "data"\n    input_param {\n        shape: { dim: 1 dim: 3 dim: 224 dim: 224 }\n    }\n}']
    cur = 'data'
    for layer in topology:
        n = layer[1]
        if layer[0] == 'conv':
            out.append('layer {\n    bottom: "%s"\n    top: "%s"\n    name: "%s"\n    type: "Convolution"\n    convolution_param {\n'
                       '        num_output: %d\n        pad: 1\n        kernel_size: 3\n    }\n}' % (cur, n, n, layer[3]))
            out.append('layer {\n    bottom: "%s"\n    top: "%s"\n    name: "relu%s"\n    type: "ReLU"\n}' % (n, n, n[4:] if n.startswith('conv') else '_' + n))
        else:
            out.append('layer {\n    bottom: "%s"\n    top: "%s"\n    name: "%s"\n    type: "Pooling"\n    pooling_param {\n'
                       '        pool: MAX\n        kernel_size: 2\n        stride: 2\n    }\n}' % (cur, n, n))
        cur = n
    return '\n'.join(out) + '\n'
