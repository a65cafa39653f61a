VARIANTS = {"generic": {"conv3": 0, "c3flags": 0}, "conv3": {"conv3": 1, "c3flags": 0}, "conv3 skew": {"conv3": 1, "c3flags": 4},
            "conv3 skew+prio": {"conv3": 1, "c3flags": 6}}
