# The reference ships an empty vec2wav/config.py; kept so `import config` keeps working.
