"""RMI -- drop-in for the reference's recursive-model index (reference SMEM/RMI.py:4).

Same staged structure: `experts + [1]` levels of single-feature linear models, a point is
routed to expert min(scale-1, max(0, int(p))) of the next level, empty experts alias the root
model.  The reference fits with scikit-learn; here each model is the closed-form simple
least-squares line (numpy, float64), so the package has no sklearn dependency.  Predictions of
a model set imported from the reference (coefficients as plain arrays) are bit-identical to
the reference's (`coef * x + intercept`, one rounding each); predictions of a locally fitted
model differ in the last bits from sklearn's solver, which only moves the last-mile search's
starting row.
"""
import numpy as np


class LinearModel:
    """y = coef_ * x + intercept_  (what sklearn's LinearRegression holds for one feature)."""

    def __init__(self, coef=0.0, intercept=0.0):
        self.coef_ = np.asarray([float(coef)])
        self.intercept_ = float(intercept)

    def fit(self, x, y):
        x = np.asarray(x, np.float64).reshape(-1)
        y = np.asarray(y, np.float64).reshape(-1)
        xm, ym = x.mean(), y.mean()
        dx = x - xm
        var = float(np.dot(dx, dx))
        slope = float(np.dot(dx, y - ym)) / var if var > 0.0 else 0.0
        self.coef_ = np.asarray([slope])
        self.intercept_ = float(ym - slope * xm)
        return self

    def predict(self, x):
        x = np.asarray(x, np.float64).reshape(-1)
        return x * self.coef_[0] + self.intercept_


class RMI:

    def __init__(self, experts=None):
        self.experts = [100, 1000] if experts is None else list(experts)
        self.models = []

    def fit(self, x, y):
        """RMI.py:10-50, vectorised: per level, per non-empty bucket, fit the line to the
        bucket-budget target (:29-39) and route the points to the next level (:42-46)."""
        x = np.asarray(x, np.float64).reshape(-1)
        y = np.asarray(y, np.float64).reshape(-1)
        levels = self.experts + [1]
        self.models = []
        assign = np.zeros(len(x), np.int64)           # bucket of every point at the current level
        n_buckets = 1
        for scale in levels:
            level = []
            nxt = np.zeros(len(x), np.int64)
            order = np.argsort(assign, kind="stable")
            bounds = np.searchsorted(assign[order], np.arange(n_buckets + 1))
            allocated = 0.0
            for b in range(n_buckets):
                refs = order[bounds[b]:bounds[b + 1]]
                if len(refs) == 0:
                    level.append(self.models[0][0])                    # :24-26 alias the root model
                    continue
                cx, cy = x[refs], y[refs]
                if scale == 1:
                    target = cy
                else:
                    span = cy.max() - cy.min()
                    if span == 0:
                        budget = 1
                    else:
                        cy = (cy - cy.min()) / span
                        budget = len(refs) * scale / len(x)
                    target = cy * budget + allocated
                    allocated += budget
                model = LinearModel().fit(cx, target)
                level.append(model)
                p = model.predict(cx)
                nxt[refs] = np.minimum(scale - 1, np.maximum(0, p.astype(np.int64)))   # int() truncates
            self.models.append(level)
            assign = nxt
            n_buckets = scale
        return self

    def predict(self, x):
        """RMI.py:52-69."""
        x = np.asarray(x, np.float64).reshape(-1)
        idx = np.zeros(len(x), np.int64)
        result = np.zeros(len(x))
        for scale, level in zip(self.experts + [1], self.models):
            coef = np.asarray([m.coef_[0] for m in level])
            icpt = np.asarray([m.intercept_ for m in level])
            result = x * coef[idx] + icpt[idx]
            with np.errstate(invalid="ignore"):
                idx = np.minimum(scale - 1, np.maximum(0, np.trunc(result))).astype(np.int64)
        return result

    # plain-array import/export (used instead of the reference's pickles)
    def coefficients(self):
        return ([np.asarray([m.coef_[0] for m in lvl], np.float64) for lvl in self.models],
                [np.asarray([m.intercept_ for m in lvl], np.float64) for lvl in self.models])

    @classmethod
    def from_coefficients(cls, experts, coefs, icpts):
        self = cls(list(experts))
        self.models = [[LinearModel(c, i) for c, i in zip(cl, il)] for cl, il in zip(coefs, icpts)]
        return self
