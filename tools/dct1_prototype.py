"""Numerical prototype for DESIGN.md lead 1: the DFT of a real-even sequence of length N = 2M (CheapTrick's lifter
transforms, the first transform of a minimum-phase chain) through ONE real FFT of length M — a quarter-size complex FFT
on the device — plus a running sum for the odd bins (the classical DCT-I reduction):

    y_j = (f_j + f_{M-j}) / 2 - sin(pi j / M) (f_j - f_{M-j}),  j < M          Y = rfft(y)
    F_{2k} = Re Y_k        F_1 = (f_0 - f_M) / 2 + sum_j f_j cos(pi j / M)        F_{2k+1} = F_{2k-1} - Im Y_k
    X_k = 2 F_k

Prints the largest error against extended-precision cosine sums, next to the error of the half-size path in use
(np.fft.rfft of the mirrored sequence), on log-spectrum-like inputs.  CPU only."""
import numpy as np


def even_dft(f):
    m = len(f) - 1
    j = np.arange(m)
    y = 0.5 * (f[:m] + f[m - j]) - np.sin(np.pi * j / m) * (f[:m] - f[m - j])
    spec = np.fft.rfft(y)
    out = np.zeros(m + 1)
    out[0::2] = spec.real
    f1 = 0.5 * (f[0] - f[m]) + np.sum(f[1:m] * np.cos(np.pi * np.arange(1, m) / m))
    out[1::2] = f1 + np.concatenate([[0.0], np.cumsum(-spec.imag[1:m // 2])])
    return 2 * out


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for m in (512, 1024, 2048):
        e_new, e_old = [], []
        for _ in range(30):
            k = np.arange(m + 1)
            f = -8 + 4 * np.cos(np.pi * k / m * rng.uniform(1, 6)) + rng.standard_normal(m + 1) * rng.uniform(0.01, 1.0)
            x = np.concatenate([f, f[m - 1:0:-1]])
            cos = np.cos(2 * np.pi * np.outer(np.arange(m + 1), np.arange(2 * m)).astype(np.longdouble) / (2 * m))
            ref = cos @ x.astype(np.longdouble)
            scale = np.max(np.abs(ref))
            e_new.append(float(np.max(np.abs(even_dft(f) - ref)) / scale))
            e_old.append(float(np.max(np.abs(np.fft.rfft(x).real - ref)) / scale))
        print("N = %4d: real FFT of length N/2 + running sum %.1e   half-size path in use %.1e   (max error / largest coefficient)"
              % (2 * m, max(e_new), max(e_old)))
