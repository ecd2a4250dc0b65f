"""Fused SGD over the parameter arenas (pretrain.py:262-272,451: SGD, momentum 0.9, weight decay applied to
every tensor incl. BN and biases, one lr for all groups).

One launch per encoder updates master weights, momentum and the bf16 compute copy; the class derives from
torch.optim.Optimizer only so that torch LR schedulers (MultiStepLR, pretrain.py:328) drive `param_groups`."""
import torch

from . import ops


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.03, momentum=0.9, weight_decay=0.0, stores=None, grad_sync=None):
        if stores is None:
            raise ValueError('dualvar_amd.optim.SGD updates ParamStore arenas: pass stores=model.stores()')
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.stores = list(stores)
        self.grad_sync = grad_sync
        self._buf = {}

    def _momentum_buf(self, st):
        b = self._buf.get(id(st))
        if b is None or b.numel() != st.total or b.device != st.master.device:
            b = torch.zeros(st.total, dtype=torch.float32, device=st.master.device)
            self._buf[id(st)] = b
        return b

    def zero_grad(self, set_to_none=False):
        for st in self.stores:
            st.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        lr, mu, wd = float(g['lr']), float(g['momentum']), float(g['weight_decay'])
        for st in self.stores:
            if st.master is None:
                continue
            scale = 1.0
            if self.grad_sync is not None:
                scale = self.grad_sync(st)
            copy = st.cc if st.dtype != ops.DV_F32 else None
            buf = self._momentum_buf(st)
            es = ops.ESIZE[st.dtype]
            # one launch over the whole arena, or one per contiguous run of trainable tensors when part of the model is
            # frozen (classifier.py:240-246 '--train_what last': requires_grad = False on the backbone -- torch's SGD
            # would not touch those tensors, not even with weight decay)
            for a, n in st.trainable_ranges():
                ops.call('dv_sgd_momentum', st.master.data_ptr() + 4 * a, st.grad.data_ptr() + 4 * a, buf.data_ptr() + 4 * a, n,
                         lr, mu, wd, scale, st.dtype, (copy.data_ptr() + es * a) if copy is not None else None)
            st.mark_dirty(cast_done=True)
            st.pending_backward = 0           # a forward whose result was never back-propagated must not stall the overlap

    def _momentum_views(self):
        """[(index in torch's flat parameter order, momentum view shaped like the parameter)] for every parameter that
        lives in one of the arenas"""
        where = {}
        for st in self.stores:
            if st.master is None:
                continue
            buf = self._momentum_buf(st)
            for s in st.slots:
                where[id(s.tensor)] = st._view(buf, s)
        out, i = [], 0
        for g in self.param_groups:
            for p in g['params']:
                if id(p) in where:
                    out.append((i, where[id(p)]))
                i += 1
        return out

    def state_dict(self):
        """torch.optim.SGD's format (pretrain.py:343-349 stores it in the checkpoint; `--resume` feeds it back):
        param_groups with flat parameter indices, state[i]['momentum_buffer'] shaped like parameter i."""
        groups, i = [], 0
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != 'params'}
            d['params'] = list(range(i, i + len(g['params'])))
            i += len(g['params'])
            groups.append(d)
        state = {i: {'momentum_buffer': v.detach().cpu().clone()} for i, v in self._momentum_views()}
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        """Accepts torch.optim.SGD state (the reference's checkpoints) and this class's own; returns the number of
        momentum buffers restored and warns when some are missing (a silent cold restart of the momentum is a
        different training run)."""
        import warnings
        views = self._momentum_views()
        live = [st for st in self.stores if st.master is not None]
        # validate EVERYTHING first: a mismatch must leave this optimizer exactly as it was (pretrain.py logs 'optimizer state
        # not restored' and trains on -- with a half-restored momentum that would be a different, silent run)
        if len(sd['param_groups']) != len(self.param_groups):
            raise ValueError('optimizer state has %d param_groups, this optimizer %d' % (len(sd['param_groups']), len(self.param_groups)))
        if 'momentum_arenas' in sd:                         # round-1 format: one flat arena per store
            if len(sd['momentum_arenas']) != len(live) or any(b.numel() != st.total for st, b in zip(live, sd['momentum_arenas'])):
                raise ValueError('momentum arenas do not match the parameter stores')
        else:
            state = sd.get('state', {})
            staged = []
            for i, v in views:
                e = state.get(i, state.get(str(i)))
                mb = None if e is None else e.get('momentum_buffer')
                if mb is not None and tuple(mb.shape) != tuple(v.shape):
                    raise ValueError('momentum_buffer %d has shape %s, parameter has %s' % (i, tuple(mb.shape), tuple(v.shape)))
                staged.append((v, mb))
        # ... then mutate
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update({k: v for k, v in s.items() if k != 'params'})
        if 'momentum_arenas' in sd:
            for st, b in zip(live, sd['momentum_arenas']):
                self._momentum_buf(st).copy_(b)
            return len(views)
        restored = 0
        with torch.no_grad():
            for v, mb in staged:
                if mb is None:
                    v.zero_()
                else:
                    v.copy_(mb.to(device=v.device, dtype=v.dtype))
                    restored += 1
        if restored != len(views):
            warnings.warn('optimizer state: %d of %d momentum buffers restored, the rest start at zero'
                          % (restored, len(views)))
        return restored
