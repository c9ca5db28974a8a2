"""Column encoders with the reference's interface (utils/field.py:7-162): ``NumericalField``
(min-max ``K*(x-Min)`` or z-score with the population std) and ``CategoricalField`` (one-hot over the
sorted distinct values).  Same attribute names (``Min/Max/K/mu/sigma``, ``dict/rev_dict``,
``data_type``) and the same ``get_data / learn / convert / reverse / dim`` protocol, so fields built
here interchange with code written against the reference.  Vectorised: no per-element Python.
The reference's "gmm" numeric encoding is not on the DDGANI path (param.json uses "minmax",
init_attn_2 uses "mean_std") and is rejected."""
from __future__ import annotations

import numpy as np


class NumericalField:
    data_type = "Numerical Data"

    def __init__(self, model="minmax", n=5):
        if model not in ("minmax", "mean_std"):
            raise ValueError(f"NumericalField: encoding {model!r} is not part of the DDGANI path")
        self.model_name = model
        self.Min = self.Max = self.K = self.mu = self.sigma = None
        self.fit_data = None
        self.learned = False

    def get_data(self, data):
        data = np.asarray(data)
        self.fit_data = data if self.fit_data is None else np.concatenate([self.fit_data, data], axis=0)

    def learn(self):
        if self.learned:
            return
        if self.model_name == "minmax":       # utils/field.py:36-39
            self.Min = np.min(self.fit_data)
            self.Max = np.max(self.fit_data)
            self.K = 1 / (self.Max - self.Min)
        else:                                  # utils/field.py:40-42 (population std)
            self.mu = np.mean(self.fit_data)
            self.sigma = np.std(self.fit_data)
        self.learned = True

    def convert(self, data):
        assert isinstance(data, np.ndarray)
        if self.model_name == "minmax":
            return self.K * (data - self.Min)
        return (data - self.mu) / self.sigma

    def reverse(self, features):
        assert isinstance(features, np.ndarray)
        if self.model_name == "minmax":
            return features / self.K + self.Min
        return features * self.sigma + self.mu

    def dim(self):
        return 1


class CategoricalField:
    data_type = "Categorical Data"

    def __init__(self, method="one-hot", noise=None):
        if method != "one-hot" or noise is not None:
            raise ValueError("CategoricalField: only noise-free one-hot encoding is on the DDGANI path")
        self.method = method
        self.noise = noise
        self.dict = {}
        self.rev_dict = {}
        self.fit_data = None
        self._values = None

    def get_data(self, data):
        data = np.asarray(data)
        self.fit_data = data if self.fit_data is None else np.concatenate([self.fit_data, data], axis=0)

    def learn(self):
        self._values = np.unique(self.fit_data)          # sorted distinct values (utils/field.py:110-113)
        self.dict = {v: i for i, v in enumerate(self._values)}
        self.rev_dict = {i: v for i, v in enumerate(self._values)}

    def codes(self, data):
        idx = np.searchsorted(self._values, data)
        if np.any(idx >= len(self._values)) or np.any(self._values[np.minimum(idx, len(self._values) - 1)] != data):
            bad = np.asarray(data)[(idx >= len(self._values)) | (self._values[np.minimum(idx, len(self._values) - 1)] != data)]
            raise KeyError(bad[0])                       # the reference's dict lookup raises KeyError too
        return idx

    def convert(self, data):
        assert isinstance(data, np.ndarray)
        idx = self.codes(data.reshape(-1))
        features = np.zeros((idx.shape[0], len(self._values)), dtype="int")
        features[np.arange(idx.shape[0]), idx] = 1
        return features

    def reverse(self, features):
        assert isinstance(features, np.ndarray)
        assert features.shape[1] == len(self._values)
        return self._values[np.argmax(features, axis=1)].reshape(-1, 1)

    def dim(self):
        return len(self._values)
