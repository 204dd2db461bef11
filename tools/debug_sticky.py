import ctypes, json, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import oracle, style_transfer2_amd as st2
from style_transfer2_amd import engine as E
from helpers import load, tiny_setup
hip = ctypes.CDLL('libamdhip64.so')
hip.hipGetErrorString.restype = ctypes.c_char_p
def peek(tag):
    e = hip.hipPeekAtLastError()
    if e:
        print('STICKY after', tag, e, hip.hipGetErrorString(e)); hip.hipGetLastError()
# wrap every engine method
for name in dir(E.Engine):
    fn = getattr(E.Engine, name)
    if callable(fn) and not name.startswith('__'):
        def mk(fn, name):
            def w(self, *a, **k):
                r = fn(self, *a, **k); peek(name); return r
            return w
        setattr(E.Engine, name, mk(fn, name))
g = load('transfer_tiny.npz')
topo, net_params, weights, content, style, init = tiny_setup(g)
params = json.loads(str(g['params_json']))
for rep in range(3):
    st = st2.StyleTransfer(st2.HipModel(net_params, topology=topo)); peek('create')
    st.set_input(init); st.set_content(content); st.set_style(style); st.reset(); st.set_weights(weights, params)
    st.optimizer_cls = st2.AdamOptimizer; st.set_step_size(10); st.reset(); st.start()
    for i in range(3):
        st.step()
    del st; peek('del')
print('done')
