"""Synthetic scenes and a stand-in model object for tests, the smoke test and the benchmark (there are no datasets or
checkpoints offline).  NOT part of the product package ``diner_amd``: nothing in it imports this."""
