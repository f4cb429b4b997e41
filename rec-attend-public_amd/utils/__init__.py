"""Counterpart of the reference's `utils` package for the part of it on the evaluation path:
`utils.postprocess` (utils/postprocess.py).  Loggers, savers, plotting and the batch iterator
are out of scope."""
