"""wandb is the reference's only logging sink (69 call sites).  Use it when installed; otherwise a
no-op stand-in with the handful of attributes the drivers touch, so they run unchanged offline."""
try:  # pragma: no cover - depends on the environment
    import wandb  # type: ignore
except Exception:  # noqa: BLE001
    class _Run:
        name = ""

        def save(self):
            pass

    class _NoWandb:
        config = {}
        run = _Run()

        def init(self, *a, **k):
            cfg = k.get("config")
            if cfg is not None:
                self.config = dict(cfg)
            return self.run

        def log(self, *a, **k):
            pass

        def watch(self, *a, **k):
            pass

        @staticmethod
        def Image(x):
            return x

    wandb = _NoWandb()
