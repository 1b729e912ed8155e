"""Drop-in counterpart of the reference's `src` package (kernels, Gram_matrice, PDEs, InverseProblems, sample_points,
solver): same module, class, method and attribute names; the arithmetic runs in libgpk.so on an MI355X."""
