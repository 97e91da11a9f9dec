"""Learning-rate multipliers for ``torch.optim.lr_scheduler.LambdaLR`` with the reference's config
keys (utils/lr_functions.py:5-137): ``lr_fct`` in {static, exponential, polynomial, cosine,
linear-warmup-polynomial, piecewise_static}, optional warm restarts (``lr_restarts`` /
``lr_restart_vals``), ``lr_params`` {power, min_lr, warmup_iters, warmup_rate, ...}."""
import bisect
import math


class LRFcts:
    def __init__(self, config: dict, lr_restart_steps: list, lr_total_steps: int):
        self.base_lr = config['learning_rate']
        self.lr_total_steps = int(lr_total_steps)
        self.lr_fct = config['lr_fct']
        self.batchwise = config.get('lr_batchwise', False)
        self.lr_params = config.get('lr_params') or {}
        restarts = list(lr_restart_steps or [])
        self.uses_restarts = len(restarts) > 0
        if 0 not in restarts:
            restarts.insert(0, 0)
        vals = [1.0]
        rv = config.get('lr_restart_vals', 1)
        if isinstance(rv, (int, float)):
            for _ in range(1, len(restarts)):
                vals.append(vals[-1] * rv)
        else:
            assert len(rv) == len(restarts) - 1, 'lr_restart_vals list must have len(lr_restarts) - 1 entries'
            vals.extend(rv)
        if self.lr_total_steps not in restarts:
            restarts.append(self.lr_total_steps)
            vals.append(0.0)
        self.lr_restarts, self.lr_restart_vals = restarts, vals
        if self.lr_fct == 'piecewise_static':
            sched = self.lr_params['piecewise_static_schedule']
            assert all(a[0] < b[0] for a, b in zip(sched, sched[1:])), 'phases must have increasing ends'
            self.piecewise = [(int(e), float(v)) for e, v in sched]

    def _segment(self, step):
        k = max(0, bisect.bisect_right(self.lr_restarts, step) - 1)
        k = min(k, len(self.lr_restarts) - 1)
        nxt = self.lr_restarts[k + 1] if k + 1 < len(self.lr_restarts) else self.lr_restarts[k] + 1
        return step - self.lr_restarts[k], max(1, nxt - self.lr_restarts[k]), self.lr_restart_vals[k]

    def _poly(self, base, cur, total):
        power = self.lr_params.get('power', 0.9) if isinstance(self.lr_params, dict) else 0.9
        min_lr = self.lr_params.get('min_lr', 0.0) if isinstance(self.lr_params, dict) else 0.0
        frac = 1.0 - cur / max(1, total - 1)
        return (base - min_lr) * max(frac, 0.0) ** power + min_lr

    def __call__(self, step: int):
        if self.lr_fct == 'piecewise_static':
            for end, v in self.piecewise:
                if step <= end:
                    return v
            return self.piecewise[-1][1]
        if self.uses_restarts:
            cur, length, base = self._segment(step)
        else:
            cur, length, base = step, self.lr_total_steps, 1.0
        if self.lr_fct == 'static':
            return base
        if self.lr_fct == 'exponential':
            gamma = self.lr_params if isinstance(self.lr_params, (int, float)) and self.lr_params else 0.98
            return base * gamma ** cur
        if self.lr_fct == 'polynomial':
            return self._poly(base, cur, length)
        if self.lr_fct == 'cosine':
            return base * 0.5 * (1.0 + math.cos(math.pi * cur / length))
        if self.lr_fct == 'linear-warmup-polynomial':
            wi, wr = self.lr_params['warmup_iters'], self.lr_params['warmup_rate']
            if step <= wi - 1:
                return 1 - (1 - (step + 1) / wi) * (1 - wr)
            return self._poly(base, cur, length)
        raise ValueError(f"Learning rate schedule '{self.lr_fct}' not recognised.")
