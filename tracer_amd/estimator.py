"""
Monte-Carlo estimator with a confidence interval for batches of rays (reference: ray_trace_utils/estimator.py:3-56): a weighted
Welford update per batch, the interval of the mean from the effective number of batches, and the loop that keeps tracing
batches until the interval is below a target.  Host bookkeeping around the device traces; same arithmetic as the reference.
"""
import numpy as N


class Estimator(object):
    def __init__(self, n_sigmas=3., relative_CI=True):
        self.mean = N.array([0.])
        self.M2 = N.array([0.])
        self.n = 0.           # samples (rays) so far
        self.n2 = 0.          # sum of the squared batch sizes
        self.n_sigmas = n_sigmas
        self.relative_CI = relative_CI

    def update(self, values, num_samples):
        """values: the estimate(s) of one batch of num_samples rays"""
        delta = values - self.mean
        self.n += num_samples
        if self.n == num_samples:             # first batch
            self.mean = num_samples * delta / self.n
            self.M2 = num_samples * delta * (values - self.mean)
        else:
            self.mean += num_samples * delta / self.n
            self.M2 += num_samples * delta * (values - self.mean)
        self.n2 += num_samples ** 2.

    def get_CI(self):
        """n_sigmas standard errors of the mean (divided by the mean when relative_CI); infinite until two batches are in"""
        if self.n == 0:
            return N.inf * N.ones(self.mean.shape)
        denom = self.n - self.n2 / self.n
        if denom <= 0:
            return N.inf * N.ones(self.mean.shape)
        stdev = N.sqrt(self.M2 / denom)
        CI = self.n_sigmas * stdev / N.sqrt(self.n ** 2 / self.n2)
        if self.relative_CI:
            CI = CI / self.mean
        CI[stdev == 0.] = 0.
        return CI


def MCRT_to_CI(fun, target_CI, num_samples, n_sigmas=3., *args, **kwargs):
    """
    Call fun(num_rays=num_samples, *args, **kwargs) -- one traced batch, returning its estimate -- until the confidence
    interval of the running mean is below target_CI.  Returns the Estimator.
    """
    verbose = kwargs.pop('verbose', False)
    estimator = Estimator(n_sigmas)
    while (estimator.get_CI() > target_CI).any():
        samples = fun(num_rays=num_samples, *args, **kwargs)
        estimator.update(samples, num_samples=num_samples)
        if verbose:
            print('Mean: %s, CI: %s -> %s' % (estimator.mean, estimator.get_CI(), target_CI))
    return estimator
