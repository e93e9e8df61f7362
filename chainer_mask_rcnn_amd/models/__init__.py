# flake8: noqa
