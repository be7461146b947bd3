"""flare/factory.py:22-62,94-113 — the decorator registry, with this build's VitaPolicy under the reference's name."""
from __future__ import annotations


class Registry:
    def __init__(self):
        self._components = {"policy": {}, "runner": {}}

    def register_policy(self, name: str):
        def _register(cls):
            self._components["policy"][name] = cls
            return cls
        return _register

    def register_runner(self, name: str):
        def _register(cls):
            self._components["runner"][name] = cls
            return cls
        return _register

    def get_policy(self, name: str):
        if name not in self._components["policy"]:
            raise ValueError(f"Unknown policy: {name}")          # factory.py:41-42
        return self._components["policy"][name]

    def get_runner(self, name: str):
        if name not in self._components["runner"]:
            raise ValueError(f"Unknown runner: {name}")
        return self._components["runner"][name]


registry = Registry()


def _register_builtin():
    from .policy import VitaPolicy
    from .trainer import PolicyTrainer
    registry.register_policy("vita")(VitaPolicy)                  # vita_policy.py:36
    registry.register_runner("policy_trainer")(PolicyTrainer)     # vita.yaml:5


_register_builtin()


def get_policy_class(name: str):
    return registry.get_policy(name)


def get_runner_class(name: str):
    return registry.get_runner(name)
