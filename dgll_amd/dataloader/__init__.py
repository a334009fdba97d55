from .dataloader import DataLoader  # noqa: F401
