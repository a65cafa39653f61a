LAYERS = [  # long-K layers: the tile epilogue / transition cost vanishes, what remains is the K loop itself
    (256, 1024, 128, 14, 3, 1, 1, False), (256, 2048, 128, 14, 1, 1, 1, False), (64, 1024, 256, 14, 3, 1, 1, False),
    (256, 256, 256, 14, 3, 1, 1, False), (256, 256, 256, 14, 1, 1, 1, False),
]
VARIANTS = {"generic": {"conv3": 0, "c3flags": 0}, "conv3": {"conv3": 1, "c3flags": 4}}
