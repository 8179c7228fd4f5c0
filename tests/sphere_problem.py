"""The config-1 instance: data and callbacks of the reference's test/problems/sphere_regression.jl:9-31
(3 parameters, 4 residuals, 1 nonlinear equality, 1 linear equality, bounds on all variables), restated
in NumPy so the restated outer iteration can be driven without Julia."""
import numpy as np

x_l = np.array([-2.0, -1.5, 0.0])
x_u = np.array([2.0, 1.5, 2.0])
A = np.array([[1.0, 2.0, -1.0]])
b = np.array([0.5])
x0 = np.array([1.0, 0.5, 1.5])


def r(x):
    return np.array([x[0] ** 2 + x[1] ** 2 - 2 * x[0] + np.sin(x[0] + x[1]) - 1.5,
                     x[0] * x[1] + 0.5 * np.cos(2 * x[0]) - 0.8,
                     (x[0] - 1.0) ** 2 + (x[1] - 0.5) ** 2 - x[2],
                     x[2] ** 2 - x[0] + 0.3 * np.sin(x[2]) - 0.2])


def jac_r(x):
    return np.array([[2 * x[0] - 2 + np.cos(x[0] + x[1]), 2 * x[1] + np.cos(x[0] + x[1]), 0.0],
                     [x[1] - np.sin(2 * x[0]), x[0], 0.0],
                     [2 * (x[0] - 1), 2 * (x[1] - 0.5), -1.0],
                     [-1.0, 0.0, 2 * x[2] + 0.3 * np.cos(x[2])]])


def c(x):
    return np.array([x[0] ** 2 + x[1] ** 2 + x[2] ** 2 - 3])


def jac_c(x):
    return np.array([[2 * x[0], 2 * x[1], 2 * x[2]]])
