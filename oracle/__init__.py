"""CPU oracle = test infrastructure (see grape_oracle.py header). Never imported by the product path."""
