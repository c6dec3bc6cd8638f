import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from c8p_testlib import *
from gr_ieee80211_amd import engine
v = np.load(GOLD + '/viterbi_vectors.npz')
eng = engine.Engine(keep_bits=True)
for k in [0, 3, 6, 9]:
    llr = v['vit%d_llr' % k]; T = int(v['vit%d_T' % k]); cr = int(v['vit%d_cr' % k]); bits = v['vit%d_bits' % k]
    tags = np.zeros(1, abi.PKT_TAGS_DT)
    tags[0]['status'] = 0; tags[0]['cr'] = cr; tags[0]['trellis'] = T; tags[0]['total'] = len(llr); tags[0]['len'] = 10
    eng.decode(tags, llr); eng.sync()
    outs = eng.outs(); sb = eng.fetch(abi.BUF_SBITS, np.uint8)[:T]
    bad = np.nonzero(sb != bits)[0]
    print(k, 'cr', cr, 'T', T, 'nbad', len(bad), 'first', bad[:20], 'last', bad[-5:] if len(bad) else None)
