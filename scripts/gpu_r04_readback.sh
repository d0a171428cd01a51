#!/bin/bash
./scripts/readback_latency; echo ==== after 20 GB; ./scripts/readback_latency big
