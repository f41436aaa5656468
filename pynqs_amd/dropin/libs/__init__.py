"""`libs` package shim: put `<repo>/pynqs_amd/dropin` on sys.path ahead of a PyNQS checkout and
`from libs.C_extension import ...` resolves to the MI355X engine."""
