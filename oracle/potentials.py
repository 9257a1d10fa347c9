"""Closed-form potentials (negative log densities) as plain torch callables.  TEST INFRASTRUCTURE.

`sum_squares` is the reference's README/test potential (README.md:45-46, test/util.py:4-5).
The reference's other potentials live in the absent third-party package `potentials`
(imported at nfmc/sample.py:17); `funnel` is defined by the build.
All take (n, d) and return (n,).
"""
import torch


def sum_squares(x):
    """U(x) = sum_j x_j^2  (target N(0, I/2))."""
    return torch.sum(x ** 2, dim=-1)


def quadratic(a, b):
    """U(x) = sum_j a_j (x_j - b_j)^2."""
    a = torch.as_tensor(a, dtype=torch.float32)
    b = torch.as_tensor(b, dtype=torch.float32)

    def u(x):
        return torch.sum(a * (x - b) ** 2, dim=-1)

    return u


def funnel(scale: float = 3.0):
    """U(x) = x_0^2 / (2 s^2) + sum_{i>=1} [ x_i^2 / (2 e^{x_0}) + x_0 / 2 ]."""

    def u(x):
        x0 = x[:, 0]
        rest = x[:, 1:]
        return x0 ** 2 / (2 * scale ** 2) + torch.sum(rest ** 2, dim=-1) * 0.5 * torch.exp(-x0) \
            + 0.5 * (x.shape[1] - 1) * x0

    return u
