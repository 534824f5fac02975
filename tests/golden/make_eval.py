"""Constants shared by make_fixtures.py::eval_fixture and the tests."""
EVAL_T = 7  # prefix lengths 1..6 of the case tokens
